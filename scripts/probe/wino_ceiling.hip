// Upper bound of a Winograd F(2x2, 3x3) (in x, y; direct in z) form of the dominant conv family on gfx950, measured BEFORE building it
// (round-4 verdict, item 2c: "record the number whichever way it falls").  Layer L5 of unet3_a: 32 -> 64 channels, 40 x 40 x 16 per patch, 75
// patches, split-fp16 arithmetic (3 MFMA products per fp32 product) -- 0.47 ms per launch with the shipped direct kernel
// (conv3_split_kernel<true, 2, false, false, false>, profiles/r04_microbench.txt).
//
// What runs here is the transformed-domain GEMM phase with exactly the operand traffic the real kernel would have, and NOTHING else of it
// (no input transform, no fp16 split, no epilogue arithmetic: whatever this measures is a floor for the real kernel):
//   * a workgroup = 4 waves = a block of 8 Winograd tiles (4 x 8 outputs x 16 z) x 64 output channels; wave = (4 tiles) x (32 channels: NT = 2);
//   * the 16 transform positions p = (xi, nu) are processed one after the other (accumulators for all 16 would need 4 x the registers); per xi
//     row the transformed input of the block, V[tile 8][nu 4][z 16][ci 32] as (hi, lo) fp16 planes = 64 KB, is "staged" in LDS (here: written
//     once with pseudo-random finite values -- STAGE=1 re-writes it per xi row from a global read of the raw tile with a few VALU per value,
//     a cheap stand-in for the transform + split);
//   * per p a wave reads its 4 tiles' B fragments once (8 x ds_read_b128), and for each dz streams the NT x 2 weight fragments of
//     U[p][dz] from global memory / L2 (1 KB per wave-load, prefetched one dz ahead) -- the z taps are NOT extra K-blocks reading shifted
//     LDS rows: the three dz products go to three accumulators that are combined by DPP row shifts (z = MFMA column = lane & 15, zero fill
//     at the patch's z border = the conv's own padding), so B traffic is a third of the direct kernel's;
//   * M_p is folded into the 2 x 2 outputs (Y += A^T M A coefficients, 0 / +-1) in registers; the outputs are stored as fp32 at the end.
// Executed MFMAs per launch: 75 x 50 blocks x 4 waves x 16 p x 72 = 17.3 M (2.33 x fewer than the direct kernel's 40.3 M).
//
//   hipcc -O3 --offload-arch=gfx950 -o wino_ceiling wino_ceiling.hip && ./wino_ceiling [stage 0|1] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TILES = 4, NUV = 4, ZB = 16, CI = 32, NT = 1, TW = 4;
constexpr int PLANE = TILES * NUV * ZB * CI * 2;          // bytes of one fp16 component plane of one xi row: 32 KB

__device__ __forceinline__ unsigned int mix(unsigned int x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned int rnd_half2(unsigned int seed) {            // two finite fp16 values of both signs, exponents around 1.0
    const unsigned int r = mix(seed);
    const unsigned int a = ((r >> 31) << 15) | ((12u + (r >> 8 & 3u)) << 10) | (r & 1023u);
    const unsigned int b = (((r >> 30) & 1u) << 15) | ((12u + (r >> 12 & 3u)) << 10) | ((r >> 16) & 1023u);
    return a | (b << 16);
}

// Y += cx * cy * M with the coefficients of A^T (2 x 4): rows i = 0: (1, 1, 1, 0), i = 1: (0, 1, -1, -1)
__device__ __forceinline__ constexpr int acoef(int i, int k) { return i == 0 ? (k < 3 ? 1 : 0) : (k == 0 ? 0 : (k == 1 ? 1 : -1)); }

#ifndef OCC
#define OCC 3
#endif
#ifndef FOLD
#define FOLD 1
#endif
template <int STAGE>
__global__ __launch_bounds__(256, OCC) void wino_gemm_phase(const u32x4* __restrict__ wts, const float* __restrict__ raw, float* __restrict__ out, int nblocks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 planes][tile 8][nu 4][z 16][ci 32] fp16
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = 0, ch = wave;                                 // every wave: the block's 4 tiles x its own 16 output channels (NT = 1)
    const int zl = lane & 15, g = lane >> 4;
    f32x4 Y[TW][NT][4];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int o = 0; o < 4; ++o) Y[t][n][o] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* wbase = reinterpret_cast<const char*>(wts);       // uniform base + 32-bit lane offset: scalar-base global loads
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // weight fragment (p, dz, channel half, nt, comp): 1 KB each
    const u32x4* wq = reinterpret_cast<const u32x4*>(wbase) + (size_t)ch * NT * 2 * 64;      // this wave's row tile (uniform)
    auto wptr = [&](int p, int dz, int n, int c) { return reinterpret_cast<const char*>(wq + ((size_t)((p * 3 + dz) * 4 * NT + n) * 2 + c) * 64) + lane16; };
#pragma unroll 1
    for (int xi = 0; xi < 4; ++xi) {
        const float cx0 = xi < 3 ? 1.f : 0.f, cx1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);       // A^T[i][xi] (xi stays a loop variable: code size, registers)
        __syncthreads();                                          // every wave is done with the previous xi row's planes
        if (STAGE || xi == 0) {
            // stand-in for "load the raw rows, transform, split, store": 64 KB of LDS written by 256 threads = 16 x 16 B per thread; STAGE = 1
            // also reads 2 rows x 10 columns x 16 z x 32 ci fp32 = 40 KB per tile-row pair from global memory and spends ~5 VALU per value
#pragma unroll 1
            for (int k = 0; k < (2 * PLANE) / (256 * 16); ++k) {
                const int slot = tid + 256 * k;
                u32x4 v;
                if (STAGE) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(raw + ((size_t)(blockIdx.x % 1024) * 4096 + (size_t)(xi * 8 + k) * 256 + tid) * 4);
                    const float s0 = r0[0] - r0[2], s1 = r0[1] + r0[2], s2 = r0[2] - r0[1], s3 = r0[1] - r0[3];
                    const auto h01 = __builtin_amdgcn_cvt_pkrtz(s0, s1), h23 = __builtin_amdgcn_cvt_pkrtz(s2, s3);
                    const auto l01 = __builtin_amdgcn_cvt_pkrtz(s0 - (float)h01[0], s1 - (float)h01[1]), l23 = __builtin_amdgcn_cvt_pkrtz(s2 - (float)h23[0], s3 - (float)h23[1]);
                    v = u32x4{__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23), __builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
                    v[0] = (v[0] & 0x83ff83ffu) | 0x30003000u; v[1] = (v[1] & 0x83ff83ffu) | 0x30003000u;      // keep the values finite and O(1) whatever `raw` holds
                    v[2] = (v[2] & 0x83ff83ffu) | 0x20002000u; v[3] = (v[3] & 0x83ff83ffu) | 0x20002000u;
                } else {
                    const unsigned int s = (blockIdx.x * 4096u + slot) * 4u;
                    v = u32x4{rnd_half2(s), rnd_half2(s + 1), rnd_half2(s + 2), rnd_half2(s + 3)};
                }
                *reinterpret_cast<u32x4*>(lds + (size_t)slot * 16) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const int p = xi * 4 + nu;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);                    // one p at a time: nothing of the next p is hoisted over this p's accumulators
            // B fragments of the wave's 4 tiles for this p: (hi, lo), 8 channels of (z = zl) per lane
            u32x4 bf[TW][2];
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    bf[t][c] = *reinterpret_cast<const u32x4*>(lds + c * PLANE + ((((tg * TW + t) * NUV + nu) * ZB + zl) * CI + g * 8) * 2);
            f32x4 M[TW][NT];
            u32x4 wf[2][NT][2];                                   // weight fragments of one z tap (double-buffered)
            constexpr int DZ[3] = {1, 0, 2};                      // centre tap first: it initialises M, the other two are shifted into it
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int c = 0; c < 2; ++c) wf[0][n][c] = *reinterpret_cast<const u32x4*>(wptr(p, DZ[0], n, c));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < 2) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int c = 0; c < 2; ++c) wf[(k + 1) & 1][n][c] = *reinterpret_cast<const u32x4*>(wptr(p, DZ[k + 1], n, c));
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                f32x4 acc[TW][NT];
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                // hi x hi, hi x lo, lo x hi (weight component, activation component), smallest terms first
                constexpr int WI[3] = {0, 0, 1}, AI[3] = {0, 1, 0};
#pragma unroll
                for (int pr = 2; pr >= 0; --pr)
#pragma unroll
                    for (int t = 0; t < TW; ++t)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[k & 1][n][WI[pr]]),
                                                                              __builtin_bit_cast(f16x8, bf[t][AI[pr]]), acc[t][n], 0, 0, 0);
                // the z taps: out[z] = c[z] + a[z - 1] + b[z + 1] with a / c / b the products of the taps dz = 0 / 1 / 2 on the UNSHIFTED input:
                // the tap's product moves one MFMA column (lane & 15 = z) down / up, zero fill at the row ends = the conv's z padding
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (k == 0) M[t][n][e] = acc[t][n][e];
                            else {
                                const int bits = __builtin_bit_cast(int, acc[t][n][e]);
                                const int sh = k == 1 ? __builtin_amdgcn_update_dpp(0, bits, 0x111, 0xf, 0xf, true)     // row_shr:1
                                                      : __builtin_amdgcn_update_dpp(0, bits, 0x101, 0xf, 0xf, true);    // row_shl:1
                                M[t][n][e] += __builtin_bit_cast(float, sh);
                            }
                        }
            }
            // fold M_p into the 2 x 2 outputs
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
#if FOLD
                            const int cj = acoef(j, nu);
                            const float cf = (i == 0 ? cx0 : cx1) * (float)cj;
                            if (cj != 0) Y[t][n][i * 2 + j] += cf * M[t][n];
#else
                            if (i == 0 && j == 0) Y[t][n][(xi + nu) & 3] += M[t][n];      // (one add per M: a floor for the fold's 2.25)
#endif
                        }
        }
    }
    // outputs: 4 x 8 x 16 z x 64 channels fp32 per block, 16 B per lane and (tile, nt, output)
    float* ob = out + (size_t)blockIdx.x * (16 * 16 * 64);
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int o = 0; o < 4; ++o)
                *reinterpret_cast<f32x4*>(ob + ((((size_t)((tg * TW + t) * 4 + o) * 8 + (ch * NT + n) * 2 + (g >> 1)) * 16 + zl) * 8 + (g & 1) * 4)) = Y[t][n][o];
}

int main(int argc, char** argv) {
    const int stage = argc > 1 ? atoi(argv[1]) : 0, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nblocks = 75 * 100;                                 // 75 patches x (40 x 40 / 16) blocks of 4 x 4 outputs (4 Winograd tiles)
    const size_t wbytes = (size_t)16 * 3 * 4 * NT * 2 * 1024;     // 384 KB of transformed, split weights
    u32x4* w; float* raw; float* out;
    CHECK(hipMalloc(&w, wbytes)); CHECK(hipMalloc(&raw, (size_t)1025 * 4096 * 16 + (1 << 20))); CHECK(hipMalloc(&out, (size_t)nblocks * 16 * 16 * 64 * 4));
    std::vector<unsigned int> hw(wbytes / 4);
    unsigned int s = 12345u;
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; const unsigned int a = ((s >> 31) << 15) | ((10u + (s >> 8 & 3u)) << 10) | (s & 1023u);
                         const unsigned int b = (((s >> 30) & 1u) << 15) | ((10u + (s >> 12 & 3u)) << 10) | ((s >> 16) & 1023u); v = a | (b << 16); }
    CHECK(hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice));
    CHECK(hipMemset(raw, 0x3c, (size_t)1025 * 4096 * 16 + (1 << 20)));
    auto k = stage ? wino_gemm_phase<1> : wino_gemm_phase<0>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PLANE));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), 2 * PLANE, 0, w, raw, out, nblocks);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), 2 * PLANE, 0, w, raw, out, nblocks);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double mfma = (double)nblocks * 4 * 16 * (TW * NT * 9);
    printf("wino GEMM phase (stage stand-in %d): %.4f ms per launch of L5's size; %.1f M MFMAs -> %.0f TFLOP/s executed (direct kernel: 0.47 ms, 40.3 M MFMAs)\n",
           stage, ms, mfma / 1e6, mfma * 16384.0 / (ms * 1e-3) / 1e12);
    return 0;
}
