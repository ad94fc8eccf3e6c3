# Bisect of the AGGRESSOR side of the co-residency hazard: the 40-line packed-fp32 victim of pk_mfma_hazard.hip runs in
# this process while ANOTHER process keeps a U-Net busy whose split conv kernels were built with one feature removed (CT_ABL bits, see
# csrc/ct_unet.hip) or selected through the environment.  A variant that leaves the victim exact names a feature the hazard needs.
#   build (container):  scripts/build_variants.sh abl1 -DCT_ABL=1 abl2 -DCT_ABL=2 abl4 -DCT_ABL=4 abl8 -DCT_ABL=8 abl16 -DCT_ABL=16 \
#                           abl32 -DCT_ABL=32 abl64 -DCT_ABL=64 abl95 -DCT_ABL=95 abl3 -DCT_ABL=3
#   run (GPU box):      bash scripts/probe/hazard_bisect.sh > gpurun_out/hazard_bisect.txt
cd $GRAFT_REPO_ROOT
V=3deecelltracker_amd/_variants
unet_loop() {   # $1 = label; remaining args: env assignments
  local label=$1; shift
  rm -f /tmp/unet_loop.log
  ( env "$@" timeout 60 python - <<'PY' > /tmp/unet_loop.log 2>&1
import importlib, torch, time
mod = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, unet3d = mod("synth"), mod("unet3d")
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
model.predict_volume_device(vol, out=out); torch.cuda.synchronize()
print("running", flush=True)
t0 = time.time()
while time.time() - t0 < 9:
    for _ in range(20): model.predict_volume_device(vol, out=out)
    torch.cuda.synchronize()
PY
  ) &
  for i in $(seq 1 80); do grep -q running /tmp/unet_loop.log 2>/dev/null && break; sleep 0.5; done
  sleep 0.5
  printf "%-58s " "$label"
  scripts/probe/pk_mfma_hazard victimonly 60
  wait
  grep -q running /tmp/unet_loop.log || { echo "   (aggressor did not start:)"; tail -3 /tmp/unet_loop.log; }
}
echo "== victim alone"; printf "%-58s " "alone"; scripts/probe/pk_mfma_hazard victimonly 60
echo "== environment-selected aggressors (default library)"
unet_loop "f16x3 family (default)"                    CT_CONV_MATH=f16x3
unet_loop "f32-input MFMA family (control, harmless)" CT_CONV_MATH=f32
unet_loop "bf16x6 family"                             CT_CONV_MATH=bf16x6
unet_loop "f16x3, first conv = f32 kernel"            CT_CONV_MATH=f16x3 CT_FIRST_F16=0
unet_loop "f16x3, no XCD remap"                       CT_CONV_MATH=f16x3 CT_CONV_XCD=0
unet_loop "f16x3, no crop-aware early exits"          CT_CONV_MATH=f16x3 CT_CONV_CROP=0
echo "== split conv kernels with one feature removed (CT_ABL), first conv: $FIRSTENV"
# (the first conv of the f16x3 family, conv_first_f16_kernel, is an aggressor of its own -- "CT_ABL=128" below still corrupts with it in
#  place -- so the matrix runs with the f32 first conv: CT_FIRST_F16=0; FIRST=1 in the environment keeps the f16 first conv)
FIRSTENV="CT_FIRST_F16=${FIRST:-0}"
for v in 128 1 2 3 4 8 16 32 64 95 256 607 1119 1631; do
  case $v in 1) d="no MFMAs";; 2) d="MFMA operands not read from LDS";; 3) d="no MFMAs, no fragment reads";; 4) d="no staging stores to LDS";;
             8) d="no global tile loads";; 16) d="no epilogue stores / maxima";; 32) d="no SGPR pinning asm";; 64) d="no weight loads";;
             95) d="none of: MFMA, fragment reads, staging stores, tile loads, epilogue, weights";;
             128) d="return at the first statement: resources only";; 256) d="return after argument fetch + tile decode, before any LDS use";;
             607) d="95 + no per-wave LDS tables";; 1119) d="95 + no threadIdx.y reads";; 1631) d="95 + no tables + no threadIdx.y";; esac
  [ -f $V/libctamd_abl$v.so ] && unet_loop "CT_ABL=$v ($d)" CTAMD_LIB=$V/libctamd_abl$v.so CT_CONV_MATH=f16x3 $FIRSTENV
done
