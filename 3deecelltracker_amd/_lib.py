"""ctypes binding of the C ABI declared in include/ctamd.h (libctamd.so, built in-tree by
csrc/Makefile / __graft_entry__.build()).  There is NO fallback: if the HIP library is missing
or a call fails, the product path raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CTAMD_LIB", _HERE / "libctamd.so"))    # CTAMD_LIB: A/B builds of the same ABI (dev)


class CtamdError(RuntimeError):
    pass


_warned_queues = False


def check_hw_queues(what: str = "the frame loop") -> bool:
    """HIP streams share 4 hardware queues per process unless GPU_MAX_HW_QUEUES (a ROCm runtime variable, read when the runtime
    initialises) says otherwise.  The MULTI-CHAIN modes (FramePipeline: one U-Net stream + `workers` match streams, each driven by its own
    host thread) are where aliasing was measured: 6.6-9.3 ms per frame depending on the order in which the process created its streams,
    6.8 ms every time with 16 queues.  FrameChain.run_sequence (the contract loop, three streams) showed no dependence
    from 4 to 32 queues (profiles/r05_conv_experiments.txt section 16) and therefore does not call this.  The package does NOT touch the host
    application's environment: this warns, once, when the variable is missing or below 8, and returns whether it was fine."""
    global _warned_queues
    try:
        ok = int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) >= 8
    except ValueError:
        ok = False
    if not ok and not _warned_queues:
        import warnings
        _warned_queues = True
        warnings.warn(f"GPU_MAX_HW_QUEUES is {os.environ.get('GPU_MAX_HW_QUEUES', 'not set')!s} (HIP default: 4 hardware queues): {what} keeps 5-6 HIP "
                      "streams busy and they will share queues, i.e. partly serialise.  Export GPU_MAX_HW_QUEUES=16 before the process makes "
                      "its first GPU call (it cannot be changed afterwards).", RuntimeWarning, stacklevel=3)
    return ok


_lib = None
MISSING: list = []   # declared in ctamd.h but absent from the built library (tests assert it is empty)

_vp, _i, _sz, _d, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_float
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every symbol of include/ctamd.h
SIGNATURES = {
    "ct_version": (_i, []),
    "ct_error_string": (C.c_char_p, [_i]),
    "ct_device_info": (_i, [_i, _ip, C.POINTER(_sz), C.c_char_p, _sz]),
    "ct_stream_create_cu_range": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "ct_stream_destroy": (_i, [_vp]),
    "ct_unet_create": (_i, [_i, _vp, _sz, _i, C.POINTER(_vp)]),
    "ct_unet_destroy": (None, [_vp]),
    "ct_unet_num_weights": (_sz, [_i]),
    "ct_unet_patch_shape": (_i, [_i, _ip]),
    "ct_unet_workspace_bytes": (_sz, [_vp, _i]),
    "ct_unet_predict_patches": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp, _vp]),
    "ct_unet_layer_dump_floats": (_sz, [_i]),
    "ct_unet_num_conv_layers": (_i, [_vp]),
    "ct_unet_layer_info": (_i, [_vp, _i, _ip, _ip, _ip, _ip]),
    "ct_unet_layer_fold_channels": (_i, [_vp, _i]),
    "ct_unet_layer_region": (_i, [_vp, _i, _vp]),
    "ct_unet_layer_tile": (_i, [_vp, _i, _vp]),
    "ct_unet_set_timing": (_i, [_vp, _i]),
    "ct_unet_get_timing": (_i, [_vp, C.POINTER(C.c_float), _ip, _i]),
    "ct_tile_plan": (_i, [_ip, _ip, _ip, _ip, _ip]),
    "ct_tile_gather_reflect": (_i, [_vp, _ip, _ip, _ip, _i, _i, _vp, _vp]),
    "ct_tile_scatter_center": (_i, [_vp, _ip, _ip, _ip, _i, _i, _vp, _vp]),
    "ct_tile_pack_crops": (_i, [_vp, _ip, _ip, _ip, _i, _i, _vp, _vp]),
    "ct_tile_unpack_crops": (_i, [_vp, _ip, _ip, _ip, _i, _i, _vp, _vp]),
    "ct_unet_predict_volume": (_i, [_vp, _vp, _ip, _ip, _i, _i, _vp, _vp, _sz, _vp]),
    "ct_normalize_points": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "ct_denormalize_points": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ct_knn_features": (_i, [_vp, _i, _i, _vp, _vp]),
    "ct_ffn_create": (_i, [_vp, _sz, _i, C.POINTER(_vp)]),
    "ct_ffn_destroy": (None, [_vp]),
    "ct_ffn_num_weights": (_sz, []),
    "ct_ffn_workspace_bytes": (_sz, [_i, _i]),
    "ct_ffn_pairgrid": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "ct_ffn_predict": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "ct_ffn_predict_workspace_bytes": (_sz, [_i]),
    "ct_greedy_workspace_bytes": (_sz, [_i, _i]),
    "ct_greedy_match": (_i, [_vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ct_prgls_workspace_bytes": (_sz, [_i, _i, _i]),
    "ct_prgls_two_ref": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _ip, _vp, _sz, _vp]),
    "ct_prgls_prepared_bytes": (_sz, [_i]),
    "ct_prgls_prepare_ref": (_i, [_vp, _i, _d, _vp, _sz, _vp]),
    "ct_prgls_two_ref_prepared": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _ip, _vp, _sz, _vp, _sz, _vp]),
    "ct_prgls_batched_workspace_bytes": (_sz, [_i, _ip, _ip, _ip]),
    "ct_prgls_two_ref_batched": (_i, [_i, _vp, _vp, _ip, _vp, _ip, _vp, _ip, _d, _d, _i, _vp, _vp, _vp, _ip, _vp, _sz, _vp]),
    "ct_prgls_legacy": (_i, [_vp, _i, _vp, _i, _vp, _d, _i, _d, _d, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ct_dist_squares": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "ct_gaussian_kernel": (_i, [_vp, _i, _vp, _i, _d, _vp, _vp]),
    "ct_estimate_posterior": (_i, [_vp, _d, _vp, _i, _vp, _i, _d, _d, _vp, _vp]),
    "ct_solve_movements": (_i, [_d, _d, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "ct_gram_apply": (_i, [_vp, _i, _vp, _i, _vp, _d, _vp]),
    "ct_match_front_batched_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ct_match_front_batched": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _sz, _vp]),
    "ct_legacy_predict_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ct_legacy_predict_pos": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ct_legacy_predict_batched_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ct_legacy_predict_pos_batched": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _d, _d, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ct_normalize_workspace_bytes": (_sz, [_ip]),
    "ct_median": (_i, [_vp, _i, _sz, _vp, _vp, _sz, _vp]),
    "ct_normalize_image": (_i, [_vp, _i, _ip, _d, _ip, _i, _i, _vp, _vp, _sz, _vp]),
    "ct_correction_workspace_bytes": (_sz, [_ip, _i]),
    "ct_accurate_correction": (_i, [_vp, _ip, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ip, _vp, _sz, _vp]),
    "ct_correction_legacy_workspace_bytes": (_sz, [_ip, _i]),
    "ct_accurate_correction_legacy": (_i, [_vp, _vp, _i, _ip, _i, _i, _d, _i, _vp, _vp, _vp, _ip, _vp, _vp, _vp, _vp, _i, _ip, _vp, _sz, _vp]),
    "ct_trim_mean": (_i, [_vp, _i, _i, _d, _vp, _vp]),
    "ct_segment_workspace_bytes": (_sz, [_ip, _i]),
    "ct_segment_centroids": (_i, [_vp, _ip, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ct_watershed_workspace_bytes": (_sz, [_ip, _i]),
    "ct_watershed_read_stage": (_i, [_vp, _ip, _i, _i, _vp, _vp]),
    "ct_watershed_segment": (_i, [_vp, _ip, _d, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ct_watershed_workspace_bytes_ex": (_sz, [_ip, _i, _i, _i]),
    "ct_watershed_segment_ex": (_i, [_vp, _ip, _d, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}


def lib():
    """Load libctamd.so (once) and attach the prototypes.  Raises CtamdError if it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise CtamdError(
                f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
                f"or make -C {_HERE / 'csrc'}).  There is no CPU fallback.")
        try:
            handle = C.CDLL(str(LIB_PATH))
        except OSError as e:  # pragma: no cover
            raise CtamdError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:          # calling it later raises AttributeError: loud, no fallback
                MISSING.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


CT_EINVAL, CT_ESHAPE, CT_EWORKSPACE, CT_ENOTCONV = -1, -2, -3, -4      # include/ctamd.h


def check(rc: int, what: str = "ctamd call"):
    if rc != 0:
        msg = lib().ct_error_string(rc)
        raise CtamdError(f"{what} failed: {rc} ({msg.decode() if msg else '?'})")


def ivec(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise CtamdError("no ROCm device visible: this package only runs on an MI355X-class GPU (no CPU fallback)")
    return torch
