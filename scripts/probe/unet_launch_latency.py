"""U-Net volume call with an EMPTY queue in front of it (sync before every call, as a dependent per-frame chain does) against back-to-back calls:
host-side launch time of the 14-kernel sequence and what the GPU waits for it."""
import importlib, sys, time
import torch
sys.path.insert(0, ".")
def mod(n): return importlib.import_module("3deecelltracker_amd." + n)
synth, unet3d = mod("synth"), mod("unet3d")
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
for _ in range(3): model.predict_volume_device(vol, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): model.predict_volume_device(vol, out=out)
torch.cuda.synchronize(); back = (time.perf_counter() - t0) / 20
host = []; wall = []; gpu = []
for _ in range(20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    model.predict_volume_device(vol, out=out)
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); wall.append(t2 - t0); gpu.append(e0.elapsed_time(e1))
import statistics as st
print(f"back-to-back {back*1e3:.3f} ms/volume | empty queue: host call {st.median(host)*1e3:.3f} ms, events {st.median(gpu):.3f} ms, wall incl. sync {st.median(wall)*1e3:.3f} ms")
