# per-layer time of the unet3_a volume path with ablated builds of ct_unet.hip (scripts/build_variants.sh <name> "-DCT_ABL=<bits>")
echo "== default"; python scripts/microbench.py unet --layers 2>/dev/null | head -3
for v in "$@"; do echo "== $v"; CTAMD_LIB=$PWD/3deecelltracker_amd/_variants/libctamd_$v.so python scripts/microbench.py unet --layers 2>/dev/null | head -3; done
