#!/bin/bash
# scripts/ab.sh "<microbench args>" variant...   -> one compact line per variant (run on the GPU box)
args=$1; shift
for v in "$@"; do
  lib=3deecelltracker_amd/_variants/libctamd_$v.so
  [ "$v" = base ] && lib=3deecelltracker_amd/libctamd.so
  echo "== $v"
  CTAMD_LIB=$PWD/$lib python scripts/microbench.py $args 2>&1 | grep -v amdgpu.ids | tail -15 | tr "\n" " " | sed "s/ms\/vol (1 launches)//g; s/  */ /g"; echo
done
