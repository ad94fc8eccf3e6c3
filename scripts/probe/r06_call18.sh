#!/bin/bash
# round 6, call 18: W and T confined to N CUs (the U-Net keeps the whole chip), with the new sweeps -- round 5 tried 32 CUs (far worse: the old watershed needed the chip)
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
for n in 0 64 96 128 192; do echo "== --mask-cus $n"; python scripts/probe/corun.py --frames 48 --mask-cus $n 2>&1 | grep -E "^unet |unet\+lcn\+ws\+match|^lcn\+ws\+match|frames per"; done > gpurun_out/r06_c18_mask.txt 2>&1
