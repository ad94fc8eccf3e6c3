import importlib, sys, time, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3deecelltracker_amd.synth"); unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
ffn_mod = importlib.import_module("3deecelltracker_amd.ffn"); tl = importlib.import_module("3deecelltracker_amd.trackerlite")
_dev = importlib.import_module("3deecelltracker_amd._dev"); _lib = importlib.import_module("3deecelltracker_amd._lib")
L = _lib.lib()
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
ffn = ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0))
vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
x, y = synth.make_point_pair(600, seed=100, box=(512, 512, 32))
xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True); yn = (y - mean) / scale
a, b, c = _dev.points_dev(xn), _dev.points_dev(yn), _dev.points_dev(xn)
def mk(first, n):
    h = C.c_void_p(); _lib.check(L.ct_stream_create_cu_range(0, first, n, C.byref(h))); return torch.cuda.ExternalStream(h.value)
def run(s1, s2, K=5, seg=True, match=True):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            if seg:
                with torch.cuda.stream(s1): model.predict_volume_device(vol, out=out)
            if match:
                with torch.cuda.stream(s2): tl.match_device(ffn, a, b, c, 3, 3)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    return dt * 1e3
d = torch.cuda.Stream(); e = torch.cuda.Stream(); hp = torch.cuda.Stream(priority=-1)
print("seg alone      %.1f ms" % run(d, e, match=False))
print("match alone    %.1f ms" % run(d, e, seg=False))
print("both, 2 streams %.1f ms" % run(d, e))
print("both, match hi-prio %.1f ms" % run(d, hp))
for nm in (8, 16, 32):
    print("both, CU split %d/%d %.1f ms" % (256 - nm, nm, run(mk(nm, 256 - nm), mk(0, nm))))
print("match alone on 16 CUs %.1f ms" % run(d, mk(0, 16), seg=False))
print("seg alone on 240 CUs %.1f ms" % run(mk(16, 240), e, match=False))
