"""Importable alias of the `3deecelltracker_amd` package (whose directory name is not an identifier):

    import ctamd
    ctamd.install_as("CellTracker")            # optional: existing `from CellTracker.x import y` keep working
    from ctamd import unet3d, trackerlite
"""
import importlib as _il
import sys as _sys
from pathlib import Path as _Path

_root = str(_Path(__file__).resolve().parent)
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_pkg = _il.import_module("3deecelltracker_amd")
install_as = _pkg.install_as
__version__ = _pkg.__version__
for _m in ("arch", "synth", "unet3d", "ffn", "track", "trackerlite", "tracker", "coord_image_transformer", "parallel", "preprocess", "segment"):
    globals()[_m] = _il.import_module(f"3deecelltracker_amd.{_m}")
