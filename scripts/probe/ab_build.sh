# A/B of two builds of libctamd.so on the GPU box: scripts/probe/libctamd_prev.so (built here from another revision / other -D flags;
# *.so files are not tracked but travel with gpurun) against the in-tree library.  Prints the PR-GLS run hash (bit-identity), the
# batched chain's time per EM iteration and the headline bench of both.
run() { python bench.py --steps 256 --no-cpu-baseline --no-realistic-pass 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"conv_stack_ms_per_volume\"])" $1; }
cp 3deecelltracker_amd/libctamd.so /tmp/new.so
for v in new prev new prev; do
  if [ $v = new ]; then cp /tmp/new.so 3deecelltracker_amd/libctamd.so; else cp scripts/probe/libctamd_prev.so 3deecelltracker_amd/libctamd.so; fi
  python scripts/probe/prgls_hash.py $PWD | tail -1
  python scripts/microbench.py batched 600 16 | tail -1; python scripts/microbench.py batched 600 1 | tail -1
  [ "${1:-}" = nobench ] || run $v
done
cp /tmp/new.so 3deecelltracker_amd/libctamd.so
