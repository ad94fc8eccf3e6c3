import sys, importlib, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
synth = importlib.import_module("3deecelltracker_amd.synth"); unet3d = importlib.import_module("3deecelltracker_amd.unet3d"); _lib = importlib.import_module("3deecelltracker_amd._lib")
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", seed=7))
for shape in ((112, 112, 12), (224, 224, 12), (200, 112, 12), (112, 230, 12), (200, 230, 20)):
    vol = torch.randn(*shape, device="cuda")
    ref = model.predict_volume_device(vol).clone()
    ws = model._workspace(_lib.lib().ct_unet_workspace_bytes(model._handle, 128))
    ws.view(torch.float32).fill_(float("nan"))
    out = model.predict_volume_device(vol)
    bad = torch.isnan(out)
    print(shape, "nan voxels:", int(bad.sum()), "equal:", bool(torch.equal(out, ref)), "first nan at", (bad.nonzero()[0].tolist() if bad.any() else None))
