// Sustained rate of the two fp16 MFMA shapes of gfx950 at the package power cap (register operands only, all CUs busy):
//   mfma_shapes <shape 0|1> [seconds]     0 = v_mfma_f32_16x16x32_f16 (the conv kernels' shape), 1 = v_mfma_f32_32x32x16_f16
// prints executed TFLOP/s over the interval.  Same flops per cycle on paper; the 32x32 shape reads half the operand registers per flop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x % 7 + i)); }
    if constexpr (SHAPE == 0) {
        f32x4 acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += acc[k][0];
        if (s == 12345.f) out[0] = s;
    } else {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += acc[k][0];
        if (s == 12345.f) out[0] = s;
    }
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 4.0;
    float* out; if (hipMalloc(&out, 4) != hipSuccess) { printf("no device\n"); return 2; }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;          // 8 x 4 waves per CU = 8 waves per SIMD
    const int iters = 20000;
    const double flop_per_launch = (double)blocks * 4 /*waves*/ * iters * (shape == 0 ? 8 * 16384.0 : 4 * 32768.0);
    auto t0 = std::chrono::steady_clock::now(); long launches = 0;
    for (;;) {
        if (shape == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        ++launches;
        if (launches % 4 == 0) {
            hipDeviceSynchronize();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > secs) break;
        }
    }
    hipDeviceSynchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("shape %s: %.0f TFLOP/s executed over %.1f s (%ld launches)\n", shape == 0 ? "16x16x32_f16" : "32x32x16_f16", flop_per_launch * launches / dt / 1e12, dt, launches);
    return 0;
}
