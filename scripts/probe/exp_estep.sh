# A/B of E-step builds in the pipelined benchmark and alone (same box, same call): default vs _variants/libctamd_<name>.so
b() { python bench.py --steps 128 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo default; b; b; python scripts/microbench.py batched 600 16 2>/dev/null | tail -2
for v in "$@"; do
  echo $v; export CTAMD_LIB=$PWD/3deecelltracker_amd/_variants/libctamd_$v.so; b; b; python scripts/microbench.py batched 600 16 2>/dev/null | tail -2; unset CTAMD_LIB
done
