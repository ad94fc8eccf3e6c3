cd $GRAFT_REPO_ROOT
bash scripts/prof.sh bench_r02 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-realistic-pass | head -3
timeout 900 python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; tail -c 200 gpurun_out/bench_r02.err
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_r02_k20.json
python -c "
import json
for f in ('bench_r02','bench_r02_k20'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['config']['with_discriminating_ffn']['volumes_per_s'], d['config']['match_chains_in_flight'], d['config']['frames_per_match_chain'])"
python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -1
