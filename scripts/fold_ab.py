"""A/B of the upsample-fold decoder kernels: run under CT_CONV_FOLD=0 and =1 (env read at model creation)."""
import importlib, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for fold in ("0", "1"):
    env = dict(os.environ, CT_CONV_FOLD=fold)
    out = subprocess.run([sys.executable, os.path.join(here, "quick_unet_bench.py")], env=env, capture_output=True, text=True)
    print(f"CT_CONV_FOLD={fold}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:])
