#!/bin/bash
# Copy what scripts/evidence.sh left under gpurun_out/ into profiles/ under the names DESIGN.md cites:  bash scripts/collect_evidence.sh r05
set -eu
R=${1:-r05}
cd "$(dirname "$0")/.."
P=gpurun_out/prof
cp $P/bench_${R}_kernel_stats.csv      profiles/${R}_bench_kernel_stats.csv
cp $P/unet_${R}_kernel_stats.csv       profiles/${R}_unet_fullchip_kernel_stats.csv
cp $P/unet_${R}_FETCH_SIZE.csv         profiles/${R}_unet_pmc_FETCH_SIZE.csv
cp $P/unet_${R}_WRITE_SIZE.csv         profiles/${R}_unet_pmc_WRITE_SIZE.csv
cp $P/${R}_unet_hbm_traffic.json       profiles/${R}_unet_hbm_traffic.json
cp $P/${R}_unet_sq_summary.json        profiles/${R}_unet_sq_summary.json
cp $P/watershed_${R}_kernel_stats.csv  profiles/${R}_watershed_kernel_stats.csv
cp $P/lcn_${R}_kernel_stats.csv        profiles/${R}_lcn_kernel_stats.csv
cp $P/frame_${R}_kernel_stats.csv      profiles/${R}_frame_kernel_stats.csv
cp $P/m2000_${R}_kernel_stats.csv      profiles/${R}_match2000_kernel_stats.csv
cp $P/match600_${R}_kernel_stats.csv   profiles/${R}_match600_kernel_stats.csv
cp $P/batched_${R}_kernel_stats.csv    profiles/${R}_match600_batched_kernel_stats.csv
cp gpurun_out/microbench_${R}.txt      profiles/${R}_microbench.txt
cp gpurun_out/evidence_${R}.log        profiles/${R}_evidence_log.txt
tail -n 1 gpurun_out/bench_${R}.json > profiles/${R}_bench_line.json
tail -n 1 gpurun_out/bench_${R}_steps20.json > profiles/${R}_bench_line_steps20.json
[ -s gpurun_out/bench_${R}_rccl_selftest.json ] && cp gpurun_out/bench_${R}_rccl_selftest.json profiles/${R}_bench_rccl_selftest.json
ls -la profiles/${R}_* | wc -l
