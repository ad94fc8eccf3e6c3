"""Accuracy of the two conv kernel families against an fp64 evaluation of the same network (test infra: uses the oracle).

Run on the GPU box.  Prints, per math mode, the worst conv-block error relative to the block's max |value| and the
probability-map max abs error -- the numbers DESIGN.md quotes for the split-bf16 kernels."""
import importlib, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from oracle import unet_ref as ur
    name = sys.argv[2]
    synth = importlib.import_module("3deecelltracker_amd.synth")
    unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
    arch = importlib.import_module("3deecelltracker_amd.arch").ARCHS[name]
    w = synth.make_unet_weights(name, seed=1)
    patch = np.random.default_rng(2).normal(size=arch.input_shape).astype(np.float32)
    ref_path = f"/tmp/conv_acc_ref_{name}.npz"
    if os.path.exists(ref_path):
        z = np.load(ref_path); want = z["want"]; collect = [z[f"l{i}"] for i in range(int(z["n"]))]
    else:
        collect = []
        want = ur.unet_forward(patch, w, arch, dtype=np.float64, collect=collect)
        np.savez(ref_path, want=want, n=len(collect), **{f"l{i}": c for i, c in enumerate(collect)})
    model = getattr(unet3d, name)().set_weights_dict(w)
    got, dump = model.predict_device(torch.from_numpy(patch[None]).cuda(), layer_dump=True)
    torch.cuda.synchronize()
    dump = dump.cpu().numpy().astype(np.float64)
    off = 0; worst = 0.0; rows = []
    for i, ref in enumerate(collect):
        mine = dump[off:off + ref.size].reshape(ref.shape); off += ref.size
        rel = float(np.abs(mine - ref).max() / max(1.0, np.abs(ref).max()))
        rows.append(f"{rel:.1e}"); worst = max(worst, rel)
    perr = float(np.abs(got[0].cpu().numpy().astype(np.float64) - want).max())
    print(f"{name} math={os.environ.get('CT_CONV_MATH', 'bf16x6')}: worst block rel err {worst:.2e}, prob map max abs err {perr:.2e}  [{' '.join(rows)}]")
else:
    for name in (sys.argv[1:] or ["unet3_a"]):
        for math in ("f32", "bf16x6"):
            env = dict(os.environ, CT_CONV_MATH=math)
            out = subprocess.run([sys.executable, __file__, "child", name], env=env, capture_output=True, text=True)
            print(out.stdout.strip() or out.stderr[-800:])
