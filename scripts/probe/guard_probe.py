"""Do the U-Net kernels write outside their buffers?  Workspace, input and output sit between 64-MB guard bands filled with a pattern."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mod = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, unet3d, _lib = mod("synth"), mod("unet3d"), mod("_lib")
G = 64 << 20
for name in (sys.argv[1:] or ["unet3_a"]):
    model = getattr(unet3d, name)().set_weights_dict(synth.make_unet_weights(name, seed=0))
    shape = (512, 512, 32) if name == "unet3_a" else ((256, 256, 24) if name == "unet3_b" else (160, 160, 96))
    nvox = shape[0] * shape[1] * shape[2]
    nb = 128
    wsb = _lib.lib().ct_unet_workspace_bytes(model._handle, nb)
    from math import prod
    centre, grid = unet3d.tile_plan(shape, model.arch.input_shape, (24, 24, 2))
    total = prod(grid); wsb = _lib.lib().ct_unet_workspace_bytes(model._handle, min(total, nb))
    big = torch.full((3 * G + wsb + 8 * nvox + 4 * G,), 0xAB, dtype=torch.uint8, device="cuda")
    o = G
    model._ws = big[o:o + wsb]; o += wsb + G
    o = (o + 255) & ~255
    vol = big[o:o + 4 * nvox].view(torch.float32).view(shape); o += 4 * nvox + G
    o = (o + 255) & ~255
    out = big[o:o + 4 * nvox].view(torch.float32).view(shape); o += 4 * nvox
    vol.copy_(torch.randn(shape, device="cuda"))
    keep = torch.ones(big.numel(), dtype=torch.bool, device="cuda")
    for t in (model._ws, vol.view(-1).view(torch.uint8), out.view(-1).view(torch.uint8)):
        a = t.data_ptr() - big.data_ptr(); keep[a:a + t.numel()] = False
    for rep in range(3):
        model.predict_volume_device(vol, out=out)
    torch.cuda.synchronize()
    touched = (big != 0xAB) & keep
    n = int(touched.sum())
    print(name, shape, "patches", total, "workspace", wsb >> 20, "MB: guard bytes overwritten:", n)
    if n:
        idx = torch.nonzero(touched).flatten()
        print("   first offsets", idx[:8].tolist(), "ws at", model._ws.data_ptr() - big.data_ptr(), "..", model._ws.data_ptr() - big.data_ptr() + wsb)
