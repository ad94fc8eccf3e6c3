"""MI355X-native hot path for 3DeeCellTracker-style tracking.

The directory name is not a Python identifier; import it with
``importlib.import_module("3deecelltracker_amd")`` or through the root-level ``ctamd`` alias module.
``install_as("CellTracker")`` registers the mirrors under the reference's import name so that
``from CellTracker.trackerlite import TrackerLite`` in an existing notebook resolves to this package.
"""
import importlib
import sys

__version__ = "0.1.0"
_MIRRORS = ("unet3d", "ffn", "track", "trackerlite", "tracker", "coord_image_transformer", "preprocess")


def install_as(name: str = "CellTracker"):
    """Alias this package (and its reference-named sub-modules) as `name` in sys.modules."""
    pkg = sys.modules[__name__]
    sys.modules[name] = pkg
    for sub in _MIRRORS:
        sys.modules[f"{name}.{sub}"] = importlib.import_module(f"{__name__}.{sub}")
    return pkg
