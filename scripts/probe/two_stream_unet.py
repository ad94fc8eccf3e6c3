"""Do two halves of a volume's patches on two streams beat one launch sequence?  (the tail of every layer's grid overlaps the other half's
body)  usage: python scripts/probe/two_stream_unet.py"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mod = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, unet3d = mod("synth"), mod("unet3d")
w = synth.make_unet_weights("unet3_a", 0)
models = [unet3d.unet3_a().set_weights_dict(w) for _ in range(4)]
vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
ref = models[0].predict_volume_device(vol).clone()
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(f"one stream, 75 patches: {timeit(lambda: models[0].predict_volume_device(vol, out=out)):.3f} ms")
for parts in (2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    bounds = [round(75 * k / parts) for k in range(parts + 1)]
    def run():
        cur = torch.cuda.current_stream()
        ev = cur.record_event()
        for k, st in enumerate(streams):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                models[k].predict_volume_device(vol, p_begin=bounds[k], n=bounds[k + 1] - bounds[k], out=out)
            cur.wait_event(st.record_event())
    ms = timeit(run)
    run(); torch.cuda.synchronize()
    print(f"{parts} streams: {ms:.3f} ms, identical: {bool(torch.equal(out, ref))}")
def seq():
    models[0].predict_volume_device(vol, p_begin=0, n=38, out=out); models[0].predict_volume_device(vol, p_begin=38, n=37, out=out)
print(f"two halves one after the other on one stream: {timeit(seq):.3f} ms")
