#!/bin/bash
# What the watershed costs a co-running U-Net as a function of the LDS floods' grid (every flood workgroup holds a CU's LDS):  bash scripts/probe/ws_grid.sh
cd "$(dirname "$0")/../.."
for g in 512 256 128 64 32 16; do
  echo "== CT_WS_FLOOD_GRID=$g"
  CT_WS_FLOOD_GRID=$g python scripts/probe/corun.py --frames 40 --only "unet+ws,ws,unet+lcn+ws+match" 2>&1 | grep -v amdgpu.ids | grep -v "frames per"
done
