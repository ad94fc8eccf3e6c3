"""Where does the device watershed leave the oracle?  Stage-by-stage comparison at the benchmark's size (GPU box)."""
import importlib, os, sys
import numpy as np, torch
import scipy.ndimage as ndi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import watershed_ref as wr
synth = importlib.import_module("3deecelltracker_amd.synth"); seg = importlib.import_module("3deecelltracker_amd.segment")
stack, _ = synth.make_stack((512, 512, 32), 600, seed=0)
prob = np.clip((stack.astype(np.float32) - 100.0) / 600.0, 0, 1)
d = torch.from_numpy(prob).cuda()
st2 = seg.watershed_stages_device(d, 4.0, "2d")
col = []
wo, bd = wr.watershed_2d(prob, prob.shape[2], 7, collect=col)
for z, c in enumerate(col):
    for key, dev in (("dist", st2["edt"]), ("dist_smooth", st2["smooth"]), ("labels", st2["labels"])):
        a, b = c[key], dev[:, :, z]
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            print(f"2D z={z} {key}: {len(bad)} differ; first {bad[0]} oracle {a[tuple(bad[0])]!r} device {b[tuple(bad[0])]!r}")
            break
    pk_o = np.argwhere(c["peaks"]); 
print("2D mask w/o boundaries equal:", np.array_equal(wo, st2["mask_wo_boundaries"].astype(bool)), int((wo != st2["mask_wo_boundaries"].astype(bool)).sum()))
st3 = seg.watershed_stages_device(d, 4.0, "3d", min_size=20)
col3 = []
wr.watershed_3d(wo, [1, 1, 4.0], "min_size", 20, 0, 3, collect=col3)
c = col3[0]
for key, dev in (("dist_smooth", st3["smooth"]), ("labels", st3["labels"])):
    a, b = c[key], dev
    eq = np.array_equal(a, b)
    print(f"3D {key} equal: {eq}")
    if not eq:
        bad = np.argwhere(a != b); print("   ", len(bad), "differ; first", bad[0], repr(a[tuple(bad[0])]), repr(b[tuple(bad[0])]))
mx = ndi.maximum_filter(c["dist_smooth"], footprint=np.ones((7, 7, 7), bool), mode="constant")
print("3D window max equal:", np.array_equal(mx, st3["window_max"]))
pk = c["peaks"]; print("oracle 3D peaks", int(pk.sum()), "device markers", int((st3["labels"] > 0).sum() > 0), "max device label", int(st3["labels"].max()), "max oracle label", int(c["labels"].max()))
cand = (c["dist_smooth"] == mx) & (c["dist_smooth"] > c["dist_smooth"].min())
print("oracle candidates before spacing", int(cand.sum()))
