// exp_nonpos (3deecelltracker_amd/csrc/ct_exp.h) against the device library's exp, bit for bit, on the GPU:
//   exp_check [count]   ->  "exp_check: <mismatches> mismatches in <count> arguments"
// Arguments: a dense random sweep of [-1100, 0] (uniform in x and uniform in log|x|), the neighbourhood of every multiple of ln2 / 2
// (where rint(x log2 e) switches), the denormal range of the result, -0.0, 0, -inf, NaN.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../3deecelltracker_amd/csrc/ct_exp.h"

__device__ uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

__global__ void check(long long count, unsigned long long* bad, double* first_bad) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t h = mix((uint64_t)i);
    const double u = (double)(h >> 11) * 0x1p-53;                    // [0, 1)
    double x;
    switch (i & 3) {
        case 0: x = -1100.0 * u; break;                              // uniform
        case 1: x = -exp2(-60.0 + 71.0 * u); break;                  // uniform in log|x|: 2^-60 .. 2^11
        case 2: { const double k = (double)((h >> 3) % 3200); const double eps = ((double)((h >> 40) & 1023) - 512.0) * 0x1p-50;
                  x = -(k * 0.34657359027997264) * (1.0 + eps); break; }          // around multiples of ln2 / 2
        default: x = -(708.0 + 40.0 * u); break;                     // denormal results and the flush to zero
    }
    if (i == 0) x = 0.0; if (i == 4) x = -0.0; if (i == 8) x = -INFINITY; if (i == 12) x = NAN; if (i == 16) x = -1075.0; if (i == 20) x = -1075.0000000000002;
    const double a = exp(x), b = exp_nonpos(x);
    if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b)) {
        if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = x; first_bad[1] = a; first_bad[2] = b; }
    }
}

int main(int argc, char** argv) {
    const long long count = argc > 1 ? atoll(argv[1]) : (1ll << 28);
    unsigned long long* bad; double* fb;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&fb, 24) != hipSuccess) { printf("exp_check: no device\n"); return 2; }
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(check, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, count, bad, fb);
    unsigned long long hb = 0; double hf[3] = {0, 0, 0};
    if (hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("exp_check: launch failed\n"); return 2; }
    hipMemcpy(hf, fb, 24, hipMemcpyDeviceToHost);
    printf("exp_check: %llu mismatches in %lld arguments\n", hb, count);
    if (hb) printf("first: x = %a  exp = %a  exp_nonpos = %a\n", hf[0], hf[1], hf[2]);
    return hb ? 1 : 0;
}
