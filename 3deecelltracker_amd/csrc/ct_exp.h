// Device-side exp for non-positive arguments (the PR-GLS E-step).  Included by ct_match.hip and by scripts/probe/exp_check.hip, which
// compares it bit for bit with the device library's exp on the GPU (tests/test_gpu_match.py::test_exp_nonpos_is_the_library_exp).
#pragma once
#include <hip/hip_runtime.h>

// exp(x) for x <= 0 (or NaN): the device library's algorithm restated operation for operation (n = rint(x log2 e); r = x - n ln2 in two
// pieces; degree-11 Horner polynomial with its coefficients; ldexp; 0 below -1075), so the value is the one `exp` returns, bit for bit.
// Why it exists: the compiler evaluates the library's Horner steps with the two-address v_fmac_f64, whose addend is the destination -- every
// step then needs a 64-bit register copy of its coefficient first (11 v_mov_b64 per exponential, a fifth of the E-step's instructions).
// v_fma_f64 in its three-address form takes the coefficient where it lives.
// (coefficient in a scalar register pair: one constant-bus operand per instruction is allowed, and the vector registers stay free)
__device__ __forceinline__ double fma_vvs(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ double fma_svv(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double exp_nonpos(double x) {
    const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma_svv(-0x1.62e42fefa39efp-1, n, x);
    r = fma_svv(-0x1.abc9e3b39803fp-56, n, r);
    double p = 0x1.ade156a5dcb37p-26;
    p = fma_vvs(r, p, 0x1.28af3fca7ab0cp-22);
    p = fma_vvs(r, p, 0x1.71dee623fde64p-19);
    p = fma_vvs(r, p, 0x1.a01997c89e6b0p-16);
    p = fma_vvs(r, p, 0x1.a01a014761f6ep-13);
    p = fma_vvs(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma_vvs(r, p, 0x1.1111111122322p-7);
    p = fma_vvs(r, p, 0x1.55555555502a1p-5);
    p = fma_vvs(r, p, 0x1.5555555555511p-3);
    p = fma_vvs(r, p, 0x1.000000000000bp-1);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    const double z = __builtin_ldexp(p, (int)n);
    return x < -1075.0 ? 0.0 : z;
}
