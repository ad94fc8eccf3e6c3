# Round evidence on the GPU box (run through gpurun): GPU tests, smoke, rocprofv3 kernel stats of the benchmark command and of
# the U-Net alone, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate PMC passes), SQ counters, the match chain, the bench line.
#   usage: bash scripts/evidence.sh r02
set -u
R=${1:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/prof.sh bench_$R $GRAFT_REPO_ROOT/bench.py --steps 64 --windows 1 --no-cpu-baseline --no-realistic-pass | head -4
bash scripts/prof.sh unet_$R $GRAFT_REPO_ROOT/scripts/microbench.py unet | head -4
bash scripts/prof_pmc.sh unet_$R FETCH_SIZE $GRAFT_REPO_ROOT/scripts/microbench.py unet | head -3
bash scripts/prof_pmc.sh unet_$R WRITE_SIZE $GRAFT_REPO_ROOT/scripts/microbench.py unet | head -3
bash scripts/prof_sq.sh unetA_$R "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $GRAFT_REPO_ROOT/scripts/microbench.py unet | tail -1
bash scripts/prof_sq.sh unetB_$R "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" $GRAFT_REPO_ROOT/scripts/microbench.py unet | tail -1
bash scripts/prof_sq.sh unetC_$R "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" $GRAFT_REPO_ROOT/scripts/microbench.py unet | tail -1
cd $GRAFT_REPO_ROOT
python scripts/sq_summary.py gpurun_out/prof/unet_${R}_kernel_stats.csv gpurun_out/prof/${R}_unet_sq_summary.json gpurun_out/prof/unetA_${R}_sq.csv gpurun_out/prof/unetB_${R}_sq.csv gpurun_out/prof/unetC_${R}_sq.csv
bash scripts/prof.sh watershed_$R $GRAFT_REPO_ROOT/scripts/microbench.py watershed | head -3
bash scripts/prof.sh lcn_$R $GRAFT_REPO_ROOT/scripts/microbench.py lcn | head -3
bash scripts/prof.sh frame_$R $GRAFT_REPO_ROOT/scripts/probe/seqonly.py | head -3      # the frame loop (FrameChain.run_sequence): the contract line's path
grep "frame sequence" /tmp/prof_frame_$R.log
bash scripts/prof.sh m2000_$R $GRAFT_REPO_ROOT/scripts/probe/m2000.py | head -3
bash scripts/prof.sh match600_$R $GRAFT_REPO_ROOT/scripts/microbench.py match 600 | head -3
bash scripts/prof.sh batched_$R $GRAFT_REPO_ROOT/scripts/microbench.py batched 600 16 | head -3
cd $GRAFT_REPO_ROOT
python scripts/hbm_traffic.py gpurun_out/prof/unet_${R}_FETCH_SIZE.csv gpurun_out/prof/unet_${R}_WRITE_SIZE.csv gpurun_out/prof/${R}_unet_hbm_traffic.json "python scripts/microbench.py unet" | head -30
for s in unet "unet --layers" lcn segment watershed correction "match 600" "goodprior 600" legacy ensemble frame pcie; do timeout 300 python scripts/microbench.py $s 2>&1 | grep -v amdgpu.ids | tail -16; done > gpurun_out/microbench_$R.txt 2>&1
tail -60 gpurun_out/microbench_$R.txt
timeout 900 python bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err; tail -c 400 gpurun_out/bench_$R.err; head -c 1500 gpurun_out/bench_$R.json
timeout 600 python bench.py --steps 20 --warmup 5 --windows 2 --no-cpu-baseline --no-realistic-pass --rccl-selftest 2>/dev/null | tail -n 1 > gpurun_out/bench_${R}_rccl_selftest.json; head -c 200 gpurun_out/bench_${R}_rccl_selftest.json; echo
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${R}_steps20.json 2>> gpurun_out/bench_$R.err; head -c 300 gpurun_out/bench_${R}_steps20.json
