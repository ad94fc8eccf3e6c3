// What the fp16 matrix pipes of gfx950 deliver, by operand content and by how long the loop has been running (DVFS):
//   mfma_ceiling <shape 0|1> <operands 0|1|2> <seconds>
//     shape     0 = v_mfma_f32_16x16x32_f16 (the conv kernels' shape), 1 = v_mfma_f32_32x32x16_f16
//     operands  0 = all-zero A and B, 1 = small ramp (the r03 probe's values), 2 = random bit patterns (finite fp16, both signs)
// Register operands only, 8 workgroups x 4 waves per CU, independent accumulators.  Every launch (~1 ms) is timed with HIP events; printed:
// executed TFLOP/s of the first 20 ms, of every 0.25-s window after that, and of the whole run.  Clock / power are sampled from outside
// (scripts/probe/mfma_ceiling.sh: rocm-smi every 0.5 s).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned int mix(unsigned int x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, int operands) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        if (operands == 0) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
        else if (operands == 1) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x % 7 + i)); }
        else {
            // random sign + 10 random mantissa bits, exponent in [-8, 7] around 1.0: finite, products never overflow fp32 in 2^31 steps
            const unsigned int r = mix((blockIdx.x * 256u + threadIdx.x) * 16u + i), s = mix(r + 0x9e3779b9u);
            const unsigned short ha = (unsigned short)(((r >> 31) << 15) | ((7u + (r >> 8 & 15u)) << 10) | (r & 1023u));
            const unsigned short hb = (unsigned short)(((s >> 31) << 15) | ((7u + (s >> 8 & 15u)) << 10) | (s & 1023u));
            a[i] = __builtin_bit_cast(_Float16, ha); b[i] = __builtin_bit_cast(_Float16, hb);
        }
    }
    // The matrix loop is inline assembly on purpose: written as builtins, hipcc (ROCm 7.2) rotates the accumulator tuples through the loop
    // (a[24:27] <- a[22:25] ...) and pays ~40 v_accvgpr_read/write copies per 8 MFMAs -- the loop then measures the copies, not the pipe
    // (that is what profiles/r03_mfma_shapes.txt's "1346 TFLOP/s" for 16x16x32 was).
    if constexpr (SHAPE == 0) {
        f32x4 c0{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n"
                         "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n"
                         "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n"
                         "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_f16 %7, %8, %9, %7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
        const float s = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];
        if (s == 12345.f) out[0] = s;
    } else {
        f32x16 c0, c1, c2, c3;
        for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        const float s = c0[0] + c1[0] + c2[0] + c3[0];
        if (s == 12345.f) out[0] = s;
    }
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 0, operands = argc > 2 ? atoi(argv[2]) : 2;
    const double secs = argc > 3 ? atof(argv[3]) : 5.0;
    float* out; if (hipMalloc(&out, 4) != hipSuccess) { printf("no device\n"); return 2; }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;
    const int iters = 2000;                                   // ~1 ms per launch at 1.3 PFLOP/s
    const double flop_per_launch = (double)blocks * 4 * iters * (shape == 0 ? 8 * 16384.0 : 4 * 32768.0);
    const int batch = 64;
    std::vector<hipEvent_t> ev(batch + 1);
    for (auto& e : ev) hipEventCreate(&e);
    std::vector<double> t_end, dur;                           // per launch: end time since start (s), duration (s)
    double now = 0.0;
    hipDeviceSynchronize();
    while (now < secs) {
        hipEventRecord(ev[0], 0);
        for (int l = 0; l < batch; ++l) {
            if (shape == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, out, iters, operands);
            else hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, out, iters, operands);
            hipEventRecord(ev[l + 1], 0);
        }
        hipEventSynchronize(ev[batch]);
        for (int l = 0; l < batch; ++l) { float ms; hipEventElapsedTime(&ms, ev[l], ev[l + 1]); now += ms * 1e-3; t_end.push_back(now); dur.push_back(ms * 1e-3); }
    }
    auto rate = [&](double a, double b) { double f = 0, t = 0; for (size_t i = 0; i < dur.size(); ++i) if (t_end[i] > a && t_end[i] <= b) { f += flop_per_launch; t += dur[i]; } return t > 0 ? f / t / 1e12 : 0.0; };
    const char* sn = shape == 0 ? "16x16x32_f16" : "32x32x16_f16";
    const char* on = operands == 0 ? "zero" : (operands == 1 ? "ramp" : "random");
    printf("shape %s operands %-6s : first 20 ms %5.0f | first 0.1 s %5.0f | whole %.1f s %5.0f TFLOP/s | windows of 0.25 s:", sn, on, rate(0, 0.02), rate(0, 0.1), now, rate(0, 1e9));
    for (double w = 0; w < now && w < 6.0; w += 0.25) printf(" %.0f", rate(w, w + 0.25));
    printf("\n");
    return 0;
}
