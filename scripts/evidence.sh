set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/prof.sh bench_r1 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-realistic-pass | head -12
bash scripts/prof.sh unet_r1 $GRAFT_REPO_ROOT/scripts/microbench.py unet | head -12
bash scripts/prof_pmc.sh unet_r1 FETCH_SIZE $GRAFT_REPO_ROOT/scripts/microbench.py unet | head -5
bash scripts/prof_pmc.sh unet_r1 WRITE_SIZE $GRAFT_REPO_ROOT/scripts/microbench.py unet | head -5
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/bench_r1_bf.json 2> gpurun_out/bench_r1_bf.err; tail -c 1500 gpurun_out/bench_r1_bf.json
