set -u
R=r04
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
bash scripts/prof.sh watershed_$R $GRAFT_REPO_ROOT/scripts/microbench.py watershed | head -3
bash scripts/prof.sh frame_$R $GRAFT_REPO_ROOT/scripts/microbench.py frame | head -3
bash scripts/prof.sh bench_$R $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-realistic-pass | head -4
cd $GRAFT_REPO_ROOT
for s in watershed frame; do timeout 300 python scripts/microbench.py $s 2>&1 | grep -v amdgpu.ids | tail -12; done > gpurun_out/microbench_tail_$R.txt 2>&1
cat gpurun_out/microbench_tail_$R.txt
timeout 900 python bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err; tail -c 300 gpurun_out/bench_$R.err; head -c 300 gpurun_out/bench_$R.json; echo
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${R}_steps20.json 2>> gpurun_out/bench_$R.err; head -c 300 gpurun_out/bench_${R}_steps20.json; echo
