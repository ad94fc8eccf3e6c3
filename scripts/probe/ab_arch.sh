#!/bin/bash
# patch throughput of the three architectures for A/B builds (scripts/build_variants.sh), interleaved twice:  ab_arch.sh shipped lb3
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = shipped ]; then unset CTAMD_LIB; else export CTAMD_LIB=$PWD/3deecelltracker_amd/_variants/libctamd_$v.so; fi
    echo "== $v (pass $rep)"; python scripts/microbench.py arch 2>&1 | grep -v amdgpu.ids
  done
done
