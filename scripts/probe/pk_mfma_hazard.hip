// Reproducer for the co-residency hazard recorded in DESIGN.md section 5 (packed-fp32 results wrong in lanes 48-63 beside the split
// conv kernels).
//   In-process attempts (two streams; none reproduces it): a packed-fp32 recurrence beside loops of v_mfma_f32_16x16x32_f16 /
//   v_mfma_f32_16x16x4_f32; a kernel with ffn_pair_kernel's structure (victim2) beside an MFMA loop fed from LDS through v_cvt_pkrtz.
//   Cross-process (scripts/probe/pk_hazard_xproc.sh; reproduces it): "victimonly N" / "victim1only N" / "victim3only N" run one victim
//   N times against its first result while another process keeps the real U-Net busy.  victim2 (LDS-fed packed fp32 with op_sel):
//   every launch wrong in target rows = 3 (mod 4); victim (packed fp32, no LDS) and victim3 (the LDS traffic, no arithmetic): exact.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/pk_mfma_hazard.hip -o scripts/probe/pk_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void aggressor(float* out, int iters, int mode) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x % 7 + i)); }
    float4v acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc2, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(0.001f * threadIdx.x, 0.002f, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.002f, 0.001f * threadIdx.x, acc2, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc2[2] + acc2[3];
}

// every lane runs the same recurrence on packed pairs; all lanes of a wave must end with the same bits
__global__ __launch_bounds__(256) void victim(float* out, int iters) {
    float2v x = {1.0f, 0.5f}, y = {0.25f, 0.125f};
    const float2v c = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; ++it) {
        x = __builtin_elementwise_fma(x, c, d);          // v_pk_fma_f32
        y = y * c;                                       // v_pk_mul_f32
        x = x + y;                                       // v_pk_add_f32
        y = y + d;
        if (x[0] > 4.f) { x *= 0.25f; y *= 0.25f; }
    }
    out[(blockIdx.x * 256 + threadIdx.x) * 2] = x[0] + y[0];
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = x[1] + y[1];
}

// victim 2: the structure of ffn_pair_kernel (LDS tiles, broadcast reads, 2 x 2 pairs per thread -> packed fp32 after SLP)
__global__ __launch_bounds__(256) void victim2(const float* __restrict__ U, const float* __restrict__ V, float* __restrict__ corr, int n) {
    constexpr int KC = 64, HID = 512;
    __shared__ float Us[32][KC + 1];
    __shared__ float Vs[32][KC + 1];
    __shared__ float inv_s[KC], mean_s[KC], beta_s[KC], w3_s[KC];
    const int tid = threadIdx.x, tr = tid & 15, tt = tid >> 4;
    const int r0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    float acc[2][2] = {};
    for (int k0 = 0; k0 < HID; k0 += KC) {
        for (int e = tid; e < 32 * KC; e += 256) {
            const int row = e / KC, kk = e - row * KC;
            Us[row][kk] = U[(size_t)(r0 + row) * HID + k0 + kk];
            Vs[row][kk] = V[(size_t)(t0 + row) * HID + k0 + kk];
        }
        if (tid < KC) { inv_s[tid] = 1.0f + 0.001f * tid; mean_s[tid] = 0.01f * tid; beta_s[tid] = 0.02f; w3_s[tid] = 0.05f - 0.001f * tid; }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < KC; ++kk) {
            const float u0 = Us[tr][kk], u1 = Us[tr + 16][kk];
            const float v0 = Vs[tt][kk], v1 = Vs[tt + 16][kk];
            const float inv = inv_s[kk], mu = mean_s[kk], be = beta_s[kk], ww = w3_s[kk];
            float y;
            y = ((u0 + v0) - mu) * inv + be; y = y >= 0.f ? y : y * 0.3f; acc[0][0] = fmaf(y, ww, acc[0][0]);
            y = ((u1 + v0) - mu) * inv + be; y = y >= 0.f ? y : y * 0.3f; acc[0][1] = fmaf(y, ww, acc[0][1]);
            y = ((u0 + v1) - mu) * inv + be; y = y >= 0.f ? y : y * 0.3f; acc[1][0] = fmaf(y, ww, acc[1][0]);
            y = ((u1 + v1) - mu) * inv + be; y = y >= 0.f ? y : y * 0.3f; acc[1][1] = fmaf(y, ww, acc[1][1]);
        }
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) corr[(size_t)(t0 + tt + 16 * i) * n + r0 + tr + 16 * j] = acc[i][j];
}
// aggressor 2: MFMA 16x16x32 fed from LDS with fp16 conversions, like the split conv kernels' main loop
__global__ __launch_bounds__(256) void aggressor2(const float* __restrict__ in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2][8192];
    float4v acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int e = threadIdx.x; e < 4096; e += 256) {
            const float x = in[(blockIdx.x * 4096 + e + it) & 0xfffff], y2 = in[(blockIdx.x * 4096 + e + it + 7) & 0xfffff];
            auto pk = __builtin_amdgcn_cvt_pkrtz(x, y2);
            *(decltype(pk)*)&tile[it & 1][2 * e] = pk;
        }
        __syncthreads();
        for (int q = 0; q < 16; ++q) {
            const half8 a = *(const half8*)&tile[it & 1][((threadIdx.x * 8) + q * 512) & 8184];
            const half8 b = *(const half8*)&tile[it & 1][((threadIdx.x * 8) + q * 512 + 2048) & 8184];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// "victimonly N": only the pair-kernel-shaped victim, N launches compared with the first - to be run while ANOTHER PROCESS keeps the
// real U-Net busy on the same GPU (scripts/probe/pk_hazard_xproc.sh)
static int victim_only(int reps) {
    const int n = 608;
    float *U, *V, *C; hipMalloc(&U, n * 512 * 4); hipMalloc(&V, n * 512 * 4); hipMalloc(&C, n * n * 4);
    std::vector<float> h(n * 512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 2001) - 1.0f;
    hipMemcpy(U, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2246822519u) % 1777) - 0.9f;
    hipMemcpy(V, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> r2(n * n), g2(n * n);
    hipLaunchKernelGGL(victim2, dim3(n / 32, n / 32), dim3(256), 0, 0, U, V, C, n); hipDeviceSynchronize();
    hipMemcpy(r2.data(), C, r2.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, q[4] = {0, 0, 0, 0}; int badreps = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(victim2, dim3(n / 32, n / 32), dim3(256), 0, 0, U, V, C, n); hipDeviceSynchronize();
        hipMemcpy(g2.data(), C, g2.size() * 4, hipMemcpyDeviceToHost);
        long b0 = bad;
        for (size_t i = 0; i < g2.size(); ++i) if (memcmp(&g2[i], &r2[i], 4)) { ++bad; ++q[(i / n) % 4]; }
        badreps += bad != b0;
    }
    printf("victim only, %d launches: %ld differing values in %d launches; target rows mod 4: %ld %ld %ld %ld\n", reps, bad, badreps, q[0], q[1], q[2], q[3]);
    return 0;
}

// victim 3: no arithmetic at all - the pair kernel's LDS tile traffic only: stage V rows into Vs[32][65], every thread reads its two
// broadcast rows back (the ds_read pattern of the pair kernel) and XORs the bits
__global__ __launch_bounds__(256) void victim3(const float* __restrict__ V, unsigned* __restrict__ out, int n) {
    constexpr int KC = 64, HID = 512;
    __shared__ float Us[32][KC + 1];
    __shared__ float Vs[32][KC + 1];
    const int tid = threadIdx.x, tr = tid & 15, tt = tid >> 4;
    const int r0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    unsigned x0 = 0, x1 = 0;
    for (int k0 = 0; k0 < HID; k0 += KC) {
        for (int e = tid; e < 32 * KC; e += 256) {
            const int row = e / KC, kk = e - row * KC;
            Us[row][kk] = V[(size_t)(r0 + row) * HID + k0 + kk];
            Vs[row][kk] = V[(size_t)(t0 + row) * HID + k0 + kk];
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < KC; ++kk) {
            x0 ^= __float_as_uint(Us[tr][kk]) * 3u + __float_as_uint(Us[tr + 16][kk]);
            x1 ^= __float_as_uint(Vs[tt][kk]) * 5u + __float_as_uint(Vs[tt + 16][kk]);
        }
        __syncthreads();
    }
    out[((size_t)(t0 + tt) * n + r0 + tr) * 2] = x0; out[((size_t)(t0 + tt) * n + r0 + tr) * 2 + 1] = x1;
}
static int victim3_only(int reps) {
    const int n = 608;
    float* V; unsigned* C; hipMalloc(&V, n * 512 * 4); hipMalloc(&C, (size_t)n * n * 8); hipMemset(C, 0, (size_t)n * n * 8);
    std::vector<float> h(n * 512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2246822519u) % 1777) - 0.9f;
    hipMemcpy(V, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r2((size_t)n * n * 2), g2((size_t)n * n * 2);
    hipLaunchKernelGGL(victim3, dim3(n / 32, n / 32), dim3(256), 0, 0, V, C, n); hipDeviceSynchronize();
    hipMemcpy(r2.data(), C, r2.size() * 4, hipMemcpyDeviceToHost);
    long badU = 0, badV = 0, q[4] = {0, 0, 0, 0}; int badreps = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(victim3, dim3(n / 32, n / 32), dim3(256), 0, 0, V, C, n); hipDeviceSynchronize();
        hipMemcpy(g2.data(), C, g2.size() * 4, hipMemcpyDeviceToHost);
        long b0 = badU + badV;
        for (size_t i = 0; i < g2.size(); ++i) if (g2[i] != r2[i]) { if (i & 1) { ++badV; ++q[((i / 2) / n) % 4]; } else ++badU; }
        badreps += (badU + badV) != b0;
    }
    printf("LDS tile traffic only, %d launches: %ld differing Us-words, %ld differing Vs-words in %d launches; Vs by target row mod 4: %ld %ld %ld %ld\n",
           reps, badU, badV, badreps, q[0], q[1], q[2], q[3]);
    return 0;
}

// the same for the LDS-free packed-fp32 recurrence (victim 1)
static int victim1_only(int reps) {
    const int VB = 2048;
    float* vo; hipMalloc(&vo, VB * 256 * 2 * sizeof(float));
    std::vector<float> ref(VB * 256 * 2), got(VB * 256 * 2);
    hipLaunchKernelGGL(victim, dim3(VB), dim3(256), 0, 0, vo, 4000); hipDeviceSynchronize();
    hipMemcpy(ref.data(), vo, ref.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, q[4] = {0, 0, 0, 0}; int badreps = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(victim, dim3(VB), dim3(256), 0, 0, vo, 4000); hipDeviceSynchronize();
        hipMemcpy(got.data(), vo, got.size() * 4, hipMemcpyDeviceToHost);
        long b0 = bad;
        for (size_t i = 0; i < got.size(); ++i) if (memcmp(&got[i], &ref[i], 4)) { ++bad; ++q[((i / 2) & 63) / 16]; }
        badreps += bad != b0;
    }
    printf("LDS-free packed-fp32 recurrence, %d launches: %ld differing values in %d launches; lane quarters: %ld %ld %ld %ld\n", reps, bad, badreps, q[0], q[1], q[2], q[3]);
    return 0;
}


// "aggr <class> <seconds>": ONLY an aggressor, for the cross-process arrangement (scripts/probe/hazard_mfma_class.sh): waves that issue
// nothing but one class of MFMA instruction (register operands, no LDS, no memory traffic in the loop).
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));
template <int CLS>
__global__ __launch_bounds__(256) void aggr_class(float* out, int iters) {
    half8 a, b; half4 a4, b4; bf16x8v ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x % 7 + i));
                                  ab[i] = (__bf16)(0.001f * (threadIdx.x + i)); bb[i] = (__bf16)(0.002f * (threadIdx.x % 7 + i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a[i]; b4[i] = b[i]; }
    float4v acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}; float16v big = {}; int4v iacc = {0, 0, 0, 0};
    const long ai = 0x0102030405060708L + threadIdx.x, bi = 0x0101010101010101L * (threadIdx.x & 3);
    for (int it = 0; it < iters; ++it) {
        if constexpr (CLS == 0) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc2, 0, 0, 0); }
        if constexpr (CLS == 1) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(0.001f * threadIdx.x, 0.002f, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.002f, 0.001f * threadIdx.x, acc2, 0, 0, 0); }
        if constexpr (CLS == 2) { acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x16f16(b4, a4, acc2, 0, 0, 0); }
        if constexpr (CLS == 3) { big = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big, 0, 0, 0); }
        if constexpr (CLS == 4) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ab, acc2, 0, 0, 0); }
        if constexpr (CLS == 5) { big = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, big, 0, 0, 0); }
        if constexpr (CLS == 6) { iacc = __builtin_amdgcn_mfma_i32_16x16x32_i8(ai, bi, iacc, 0, 0, 0); }
        if constexpr (CLS == 7) { acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ai, bi, acc, 0, 0, 0); }
        if constexpr (CLS == 8) {             // no MFMA at all: a dependent fp32 FMA chain (control)
            acc[0] = fmaf(acc[0], 0.999f, 1e-3f); acc[1] = fmaf(acc[1], 1.001f, -1e-3f); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc2[2] + acc2[3] + big[0] + big[7] + (float)iacc[0];
}
static int aggr_only(int cls, double secs) {
    float* ao; hipMalloc(&ao, 4096 * 256 * sizeof(float));
    printf("running\n"); fflush(stdout);
    const char* names[] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_bf16",
                           "v_mfma_f32_32x32x8_f16", "v_mfma_i32_16x16x32_i8", "v_mfma_f32_16x16x32_fp8_fp8", "no MFMA (fp32 FMA chain)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    long launches = 0; float ms = 0;
    while (ms < secs * 1000.0) {
        for (int k = 0; k < 4; ++k) {
#define CT_AG(c) hipLaunchKernelGGL(aggr_class<c>, dim3(1024), dim3(256), 0, 0, ao, 20000)
            switch (cls) { case 0: CT_AG(0); break; case 1: CT_AG(1); break; case 2: CT_AG(2); break; case 3: CT_AG(3); break; case 4: CT_AG(4); break;
                           case 5: CT_AG(5); break; case 6: CT_AG(6); break; case 7: CT_AG(7); break; default: CT_AG(8); }
#undef CT_AG
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); launches += 4;
    }
    printf("aggressor %s: %ld launches of 1024 x 256 threads in %.1f s\n", names[cls < 9 ? cls : 8], launches, ms / 1000.0);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 3 && !strcmp(argv[1], "aggr")) return aggr_only(atoi(argv[2]), atof(argv[3]));
    if (argc > 2 && !strcmp(argv[1], "victimonly")) return victim_only(atoi(argv[2]));
    if (argc > 2 && !strcmp(argv[1], "victim1only")) return victim1_only(atoi(argv[2]));
    if (argc > 2 && !strcmp(argv[1], "victim3only")) return victim3_only(atoi(argv[2]));
    const int VB = 2048, AB = 512;        // 2 aggressor workgroups per CU: the victim must fit beside them
    float *vo, *ao;
    hipMalloc(&vo, VB * 256 * 2 * sizeof(float)); hipMalloc(&ao, 4096 * 256 * sizeof(float));
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<float> ref(VB * 256 * 2), got(VB * 256 * 2);
    hipLaunchKernelGGL(victim, dim3(VB), dim3(256), 0, s1, vo, 4000); hipStreamSynchronize(s1);
    hipMemcpy(ref.data(), vo, ref.size() * 4, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 2; ++mode) {
        int bad = 0, badlanes[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(aggressor, dim3(AB), dim3(256), 0, s2, ao, 400000, mode);
            hipLaunchKernelGGL(victim, dim3(VB), dim3(256), 0, s1, vo, 4000);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), vo, got.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < got.size(); ++i)
                if (memcmp(&got[i], &ref[i], 4)) { ++bad; ++badlanes[((i / 2) & 63) / 16]; }
        }
        printf("aggressor %s: %d differing values (of %zu x 5); by lane quarter: %d %d %d %d\n", mode == 0 ? "mfma 16x16x32 f16" : "mfma 16x16x4 f32",
               bad, got.size(), badlanes[0], badlanes[1], badlanes[2], badlanes[3]);
    }
    {   // the pair-kernel-shaped victim beside the LDS-fed MFMA aggressor
        const int n = 608;
        float *U, *V, *C, *in; hipMalloc(&U, n * 512 * 4); hipMalloc(&V, n * 512 * 4); hipMalloc(&C, n * n * 4); hipMalloc(&in, (1 << 20) * 4);
        std::vector<float> h(n * 512), hin(1 << 20);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 2001) - 1.0f;
        for (size_t i = 0; i < hin.size(); ++i) hin[i] = 0.001f * (float)((i * 40503u) % 1999) - 1.0f;
        hipMemcpy(U, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2246822519u) % 1777) - 0.9f;
        hipMemcpy(V, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> r2(n * n), g2(n * n);
        hipLaunchKernelGGL(victim2, dim3(n / 32, n / 32), dim3(256), 0, s1, U, V, C, n); hipDeviceSynchronize();
        hipMemcpy(r2.data(), C, r2.size() * 4, hipMemcpyDeviceToHost);
        for (int with = 0; with < 2; ++with) {
        int bad = 0, q[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 20; ++rep) {
            if (with) hipLaunchKernelGGL(aggressor2, dim3(1024), dim3(256), 0, s2, in, ao, 300);
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim2, dim3(n / 32, n / 32), dim3(256), 0, s1, U, V, C, n);
            hipDeviceSynchronize();
            hipMemcpy(g2.data(), C, g2.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < g2.size(); ++i) if (memcmp(&g2[i], &r2[i], 4)) { ++bad; ++q[(i / n) % 4]; }
        }
        printf("pair-kernel-shaped victim %s: %d differing values of %zu x 20; target rows mod 4: %d %d %d %d\n",
               with ? "beside LDS-fed mfma 16x16x32" : "alone (control)", bad, g2.size(), q[0], q[1], q[2], q[3]);
        }
    }
    return 0;
}
