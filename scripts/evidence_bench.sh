# The bench line and the rocprofv3 summary of the same command, from ONE box (so that roofline.avg_launch_ms and the kernel stats can be compared):
#   bash scripts/evidence_bench.sh r04      (through gpurun; copies nothing into profiles/ -- that is done by hand)
set -u
R=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
bash scripts/prof.sh bench_$R $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-realistic-pass | head -4
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err; tail -c 300 gpurun_out/bench_$R.err; python scripts/probe/pick.py gpurun_out/bench_$R.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${R}_steps20.json 2>> gpurun_out/bench_$R.err; python scripts/probe/pick.py gpurun_out/bench_${R}_steps20.json
