"""Does the U-Net of one volume run faster as two patch ranges on two streams (one layer's tail under the other's head)?  Same output volume
(disjoint centre crops).  python scripts/probe/unet_two_streams.py"""
import sys, time, importlib
sys.path.insert(0, '.')
import numpy as np, torch
synth = importlib.import_module('3deecelltracker_amd.synth'); unet3d = importlib.import_module('3deecelltracker_amd.unet3d')
w = synth.make_unet_weights("unet3_a", 0)
m0 = unet3d.unet3_a().set_weights_dict(w); m1 = unet3d.unet3_a().set_weights_dict(w); m2 = unet3d.unet3_a().set_weights_dict(w)
vol = torch.randn(512, 512, 32, device="cuda"); ref = torch.zeros_like(vol); out = torch.zeros_like(vol)
m0.predict_volume_device(vol, out=ref); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def single():
    m0.predict_volume_device(vol, out=ref)
def split(a):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): m1.predict_volume_device(vol, out=out, p_begin=0, n=a)
    with torch.cuda.stream(s2): m2.predict_volume_device(vol, out=out, p_begin=a, n=75 - a)
    cur.wait_stream(s1); cur.wait_stream(s2)
def timeit(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
print(f"one stream, 75 patches: {timeit(single)*1e3:.3f} ms")
for a in (38, 25, 50):
    t = timeit(lambda: split(a)); print(f"two streams, {a} + {75-a} patches: {t*1e3:.3f} ms  (identical: {bool(torch.equal(out, ref))})")
print(f"one stream again: {timeit(single)*1e3:.3f} ms")
