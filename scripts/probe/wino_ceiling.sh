#!/bin/bash
# The Winograd GEMM-phase ceiling of L5 (scripts/probe/wino_ceiling.hip) beside the shipped direct kernel's per-layer times.
cd "$(dirname "$0")"
for v in "3 1" "3 0" "2 1"; do set -- $v
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -DOCC=$1 -DFOLD=$2 -o wino_ceiling_$1_$2 wino_ceiling.hip 2>/dev/null || { echo build failed; exit 1; }
  for st in 0 1; do echo -n "OCC=$1 FOLD=$2: "; ./wino_ceiling_$1_$2 $st 50; done
done
