"""Accuracy report of the three conv arithmetic families against the fp64-accumulating oracle (a checker, hence under tests/: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg touch oracle/).  Not collected by pytest (run it directly on a GPU box):

    python tests/report_accuracy.py [unet3_a unet3_c ...]

One line per family: worst conv-block error relative to the block's scale, probability-map max abs error, per-block errors."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def mod(name):
    return importlib.import_module(f"3deecelltracker_amd.{name}")


def main(args):
    if args and args[0] == "child":
        from oracle import unet_ref as ur
        name = args[1]
        synth, unet3d, arch = mod("synth"), mod("unet3d"), mod("arch").ARCHS[name]
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
        for ref in collect:
            mine = dump[off:off + ref.size].reshape(ref.shape); off += ref.size
            rel = float(np.abs(mine - ref).max() / max(1.0, np.abs(ref).max()))
            rows.append(f"{rel:.1e}"); worst = max(worst, rel)
        perr = float(np.abs(got[0].cpu().numpy().astype(np.float64) - want).max())
        print(f"{name} math={os.environ.get('CT_CONV_MATH', 'f16x3')}: worst block rel err {worst:.2e}, prob map max abs err {perr:.2e}  [{' '.join(rows)}]")
        return
    for name in (args or ["unet3_a"]):
        for math in ("f32", "bf16x6", "f16x3"):
            out = subprocess.run([sys.executable, __file__, "child", name], env=dict(os.environ, CT_CONV_MATH=math),
                                 capture_output=True, text=True)
            print(out.stdout.strip() or out.stderr[-800:])


if __name__ == "__main__":
    main(sys.argv[1:])
