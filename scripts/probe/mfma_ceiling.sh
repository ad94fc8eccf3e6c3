#!/bin/bash
# Reconcile the guide's measured MFMA peak (MI355X_MICROARCH.md: 2495 TF, 32x32x16) with the sustained rates of r03_mfma_shapes.txt:
# operand content (zero / ramp / random bits) x loop length (0.1 s burst after an idle pause vs 5 s sustained), clock + power sampled outside.
cd "$(dirname "$0")"
[ -x ./mfma_ceiling ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o mfma_ceiling mfma_ceiling.hip || exit 1
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | sed -E 's/.*(sclk[^(]*\(([0-9]+)Mhz\)).*/sclk \2 MHz/; s/.*Power \(W\): *([0-9.]+).*/power \1 W/' | tr '\n' ' '; }
echo "idle: $(smi)"
for shape in 0 1; do for op in 0 1 2; do
    sleep 3                                          # let the package cool / clocks recover between runs
    ./mfma_ceiling $shape $op 0.1 | sed 's/| windows.*//'
    sleep 3
    ./mfma_ceiling $shape $op 5 > /tmp/mc.out &
    pid=$!
    sleep 1.5; a=$(smi); sleep 1.5; b=$(smi); sleep 1.2; c=$(smi)
    wait $pid
    cat /tmp/mc.out
    echo "        rocm-smi at 1.5 s: $a | at 3 s: $b | at 4.2 s: $c"
done; done
