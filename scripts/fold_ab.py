"""A/B of the conv kernel families (env is read at model creation): f32-input MFMA vs split-bf16, tap folding on/off."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for math, fold in (("f32", "0"), ("f32", "1"), ("bf16x6", "0"), ("bf16x6", "1")):
    env = dict(os.environ, CT_CONV_FOLD=fold, CT_CONV_MATH=math)
    out = subprocess.run([sys.executable, os.path.join(here, "quick_unet_bench.py")], env=env, capture_output=True, text=True)
    print(f"CT_CONV_MATH={math} CT_CONV_FOLD={fold}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:])
