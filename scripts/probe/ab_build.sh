# A/B of builds of libctamd.so on the GPU box.  usage: bash scripts/probe/ab_build.sh [libA.so libB.so ...]   (default: the in-tree build)
# Libraries built here from other revisions / -D flags go under scripts/probe/ (*.so is untracked but travels with gpurun) and are
# selected with CTAMD_LIB.  Prints, per library: the PR-GLS run hash (bit-identity), the batched chain's time per EM iteration, and
# the frame pipeline in its MATCH-BOUND configuration (64 match CUs: at the default 96 the U-Net stream is the longer half and hides
# what a match-side change does).
libs=${@:-3deecelltracker_amd/libctamd.so}
run() { python bench.py --match-cus 64 --match-workers 2 --match-batch 32 --steps 384 --no-cpu-baseline --no-realistic-pass 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"conv_stack_ms_per_volume\"])" $1; }
for rep in 1 2; do for l in $libs; do
  export CTAMD_LIB=$PWD/$l
  if [ $rep = 1 ]; then python scripts/probe/prgls_hash.py $PWD | tail -1; python scripts/microbench.py batched 600 16 | tail -1; fi
  run $l
done; done
