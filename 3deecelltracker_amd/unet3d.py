"""Host-side mirror of the reference's ``CellTracker/unet3d.py`` inference surface.

Same names and call contracts (reference unet3d.py:26,40,70,203):

    model = unet3_a() / unet3_b() / unet3_c()
    model.input_shape, model.output_shape, model.predict(x), model.load_weights(path)
    out = unet3_prediction(img, model, shrink=(24, 24, 2))       # numpy in, numpy float32 out

but every numeric operator runs in the hand-written HIP library (csrc/ct_unet.hip) through the C
ABI of include/ctamd.h.  torch is used for device buffers and streams only.  Training entry
points of the Keras model (compile / fit_generator / evaluate) raise NotImplementedError: training
is outside the accelerated path.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import _lib
from .arch import ARCHS, UNetArch


def flatten_unet_weights(weights: dict) -> np.ndarray:
    """dict container (see synth.py) -> the flat fp32 layout ct_unet_create expects."""
    parts = []
    for layer in weights["convs"]:
        for key in ("kernel", "bias", "gamma", "beta", "mean", "var"):
            parts.append(np.asarray(layer[key], dtype=np.float32).ravel())
    parts.append(np.asarray(weights["head"]["kernel"], dtype=np.float32).ravel())
    parts.append(np.asarray(weights["head"]["bias"], dtype=np.float32).ravel())
    return np.ascontiguousarray(np.concatenate(parts))


class UNet3Model:
    """Stand-in for the keras.Model returned by unet3_a/b/c (only what inference callers touch)."""

    def __init__(self, arch: UNetArch, device: int | None = None):
        self.arch = arch
        x, y, z = arch.input_shape
        self.input_shape = (None, x, y, z, 1)
        self.output_shape = (None, x, y, z, 1)
        self._handle = None
        self._device = device
        self._ws = None
        self._weights = None

    # ---- weights
    def set_weights_dict(self, weights: dict):
        torch = _lib.require_gpu()
        if self._device is None:
            self._device = torch.cuda.current_device()
        flat = flatten_unet_weights(weights)
        lib = _lib.lib()
        if flat.size != lib.ct_unet_num_weights(self.arch.arch_id):
            raise ValueError(f"weight count {flat.size} does not match {self.arch.name}")
        self._free()
        h = C.c_void_p()
        _lib.check(lib.ct_unet_create(self.arch.arch_id, flat.ctypes.data, flat.size, self._device, C.byref(h)),
                   "ct_unet_create")
        self._handle = h
        self._weights = weights
        return self

    def load_weights(self, path):
        """`.npz` written by save_weights, or a Keras `.h5` when h5py is importable."""
        path = Path(path)
        if path.suffix == ".npz":
            z = np.load(path)
            n = len(self.arch.conv_layers())
            convs = [{k: z[f"conv{i}_{k}"] for k in ("kernel", "bias", "gamma", "beta", "mean", "var")} for i in range(n)]
            head = {"kernel": z["head_kernel"], "bias": z["head_bias"]}
            return self.set_weights_dict({"arch": self.arch.name, "convs": convs, "head": head})
        try:
            import h5py  # noqa: F401
        except ImportError as e:
            raise OSError(f"cannot read {path}: Keras .h5 import needs h5py, which is not installed") from e
        from .keras_h5 import read_unet_h5
        return self.set_weights_dict(read_unet_h5(path, self.arch))

    def save_weights(self, path):
        if self._weights is None:
            raise ValueError("model has no weights")
        out = {}
        for i, l in enumerate(self._weights["convs"]):
            for k in ("kernel", "bias", "gamma", "beta", "mean", "var"):
                out[f"conv{i}_{k}"] = l[k]
        out["head_kernel"] = self._weights["head"]["kernel"]; out["head_bias"] = self._weights["head"]["bias"]
        np.savez(path, **out)

    # ---- inference
    def _workspace(self, nbytes: int):
        torch = _lib.require_gpu()
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{self._device}")
        return self._ws

    def _require(self):
        if self._handle is None:
            raise ValueError("model has no weights: call load_weights() or set_weights_dict() first")

    def predict_device(self, patches, layer_dump: bool = False):
        """patches: torch fp32 cuda tensor [n, X, Y, Z] -> prob [n, X, Y, Z] (device, async)."""
        torch = _lib.require_gpu()
        self._require()
        lib = _lib.lib()
        n = patches.shape[0]
        assert patches.is_cuda and patches.dtype == torch.float32 and patches.is_contiguous()
        assert tuple(patches.shape[1:]) == tuple(self.arch.input_shape)
        out = torch.empty_like(patches)
        ws = self._workspace(lib.ct_unet_workspace_bytes(self._handle, n))
        dump = None
        if layer_dump:
            dump = torch.empty(lib.ct_unet_layer_dump_floats(self.arch.arch_id), dtype=torch.float32, device=patches.device)
        st = torch.cuda.current_stream(patches.device).cuda_stream
        _lib.check(lib.ct_unet_predict_patches(self._handle, patches.data_ptr(), n, out.data_ptr(), ws.data_ptr(),
                                               ws.numel(), dump.data_ptr() if dump is not None else None, st),
                   "ct_unet_predict_patches")
        return (out, dump) if layer_dump else out

    def predict(self, x, batch_size=None, verbose=0, **_):
        """Keras-style: x numpy (n, X, Y, Z, 1) -> numpy float32 (n, X, Y, Z, 1)   (unet3d.py:253)."""
        torch = _lib.require_gpu()
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        if x.ndim != 5 or x.shape[-1] != 1 or tuple(x.shape[1:4]) != tuple(self.arch.input_shape):
            raise ValueError(f"expected input of shape (n, {self.arch.input_shape}, 1), got {x.shape}")
        dev = torch.from_numpy(x[..., 0]).to(f"cuda:{self._device if self._device is not None else torch.cuda.current_device()}")
        out = self.predict_device(dev.contiguous())
        return out.cpu().numpy()[..., None]

    def predict_volume_device(self, vol, shrink=(24, 24, 2), p_begin: int = 0, n: int | None = None, out=None,
                              max_batch: int | None = None):
        """vol: torch fp32 cuda [x, y, z] -> prob volume [x, y, z] for patches [p_begin, p_begin+n)."""
        torch = _lib.require_gpu()
        self._require()
        lib = _lib.lib()
        assert vol.is_cuda and vol.dtype == torch.float32 and vol.is_contiguous() and vol.dim() == 3
        centre, grid = tile_plan(tuple(vol.shape), self.arch.input_shape, shrink)
        total = grid[0] * grid[1] * grid[2]
        n = total - p_begin if n is None else n
        if out is None:
            out = torch.zeros_like(vol)
        if n <= 0:
            return out
        nb = min(n, 128 if max_batch is None else max_batch)        # 119 MB of workspace per patch: cap the batch
        ws = self._workspace(lib.ct_unet_workspace_bytes(self._handle, nb))
        st = torch.cuda.current_stream(vol.device).cuda_stream
        _lib.check(lib.ct_unet_predict_volume(self._handle, vol.data_ptr(), _lib.ivec(vol.shape), _lib.ivec(shrink),
                                              p_begin, n, out.data_ptr(), ws.data_ptr(), ws.numel(), st),
                   "ct_unet_predict_volume")
        return out

    # ---- training surface of the keras object: not part of the accelerated path
    def compile(self, *a, **k):
        raise NotImplementedError("training is outside the MI355X inference path")

    fit_generator = fit = evaluate = compile

    def _free(self):
        if self._handle is not None:
            _lib.lib().ct_unet_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def unet3_a(device=None) -> UNet3Model:
    """reference unet3d.py:26-37"""
    return UNet3Model(ARCHS["unet3_a"], device)


def unet3_b(device=None) -> UNet3Model:
    """reference unet3d.py:40-67"""
    return UNet3Model(ARCHS["unet3_b"], device)


def unet3_c(device=None) -> UNet3Model:
    """reference unet3d.py:70-81"""
    return UNet3Model(ARCHS["unet3_c"], device)


def load_model(path, device=None) -> UNet3Model:
    """Stand-in for keras `load_model(path)` on a U-Net file (reference tracker.py:579): the architecture is recognised from
    the stored conv shapes (`.npz` of save_weights) or, for a Keras `.h5`, by trying the three reference architectures."""
    path = Path(path)
    if not path.exists():
        raise OSError(f"Unable to open file {path}")
    if path.suffix == ".npz":
        z = np.load(path)
        shapes = []
        i = 0
        while f"conv{i}_kernel" in z.files:
            shapes.append(tuple(int(v) for v in z[f"conv{i}_kernel"].shape[3:])); i += 1
        for arch in ARCHS.values():
            if [tuple(l) for l in arch.conv_layers()] == shapes:
                return UNet3Model(arch, device).load_weights(path)
        raise ValueError(f"{path}: conv shapes {shapes} match none of {list(ARCHS)}")
    last = None
    for arch in ARCHS.values():
        try:
            return UNet3Model(arch, device).load_weights(path)
        except ValueError as e:
            last = e
    raise ValueError(f"{path}: not a weight file of {list(ARCHS)} ({last})")


def _get_sizes_padded_im(img_siz_i: int, out_centr_siz_i: int):
    """reference unet3d.py:259-279"""
    num = -(-int(img_siz_i) // int(out_centr_siz_i))
    return num * out_centr_siz_i, num


def tile_plan(vol_shape, net_shape, shrink):
    lib = _lib.lib()
    centre, grid = _lib.ivec([0, 0, 0]), _lib.ivec([0, 0, 0])
    _lib.check(lib.ct_tile_plan(_lib.ivec(vol_shape), _lib.ivec(net_shape), _lib.ivec(shrink), centre, grid), "ct_tile_plan")
    return tuple(centre), tuple(grid)


def unet3_prediction(img, model, shrink=(24, 24, 2)):
    """reference unet3d.py:203-256: img (1, x, y, z, 1) normalised -> float32 (1, x, y, z, 1).

    With a UNet3Model the whole pipeline (reflect pad, patch gather, network, centre stitch) runs on
    the GPU with no host round trips.  Any other object exposing the keras surface
    (input_shape / output_shape / predict) is driven patch by patch with the device tiler doing the
    gather / scatter, so fake models used in tests behave as in the reference."""
    torch = _lib.require_gpu()
    img = np.asarray(img)
    if img.ndim != 5:
        raise ValueError(f"img must have shape (sample, x, y, z, channel), got {img.shape}")
    on = f"cuda:{model._device}" if isinstance(model, UNet3Model) and model._device is not None else "cuda"
    vol = torch.from_numpy(np.ascontiguousarray(img[0, :, :, :, 0], dtype=np.float32)).to(on)
    if isinstance(model, UNet3Model):
        out = model.predict_volume_device(vol, shrink)
        return out.cpu().numpy()[None, :, :, :, None]
    lib = _lib.lib()
    net_in = tuple(model.input_shape[1:4]); net_out = tuple(model.output_shape[1:4])
    if net_in != net_out:
        raise ValueError("input and output patch shapes must agree")
    centre, grid = tile_plan(tuple(vol.shape), net_in, shrink)
    total = grid[0] * grid[1] * grid[2]
    out = torch.zeros_like(vol)
    st = torch.cuda.current_stream().cuda_stream
    vs, ns, sh = _lib.ivec(vol.shape), _lib.ivec(net_in), _lib.ivec(shrink)
    patch = torch.empty((1, *net_in), dtype=torch.float32, device=vol.device)
    for p in range(total):
        _lib.check(lib.ct_tile_gather_reflect(vol.data_ptr(), vs, ns, sh, p, 1, patch.data_ptr(), st), "gather")
        pred = np.asarray(model.predict(patch.cpu().numpy()[..., None]), dtype=np.float32)
        pred_d = torch.from_numpy(np.ascontiguousarray(pred[..., 0])).cuda()
        _lib.check(lib.ct_tile_scatter_center(pred_d.data_ptr(), vs, ns, sh, p, 1, out.data_ptr(), st), "scatter")
    return out.cpu().numpy()[None, :, :, :, None]
