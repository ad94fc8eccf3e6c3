// ct_match.hip -- FFN initial matching + greedy assignment + PR-GLS for gfx950 (MI355X).
//
// What it replaces (reference, numpy / sklearn / Keras):
//   CellTracker/ffn.py:268-327 (== track.py:117-178)  kNN shape features + all-pairs FFN
//   CellTracker/ffn.py:225-265                         FFN forward
//   CellTracker/trackerlite.py:242-259, track.py:58-70 greedy one-to-one assignment / prior
//   CellTracker/trackerlite.py:262-417                 PR-GLS (TrackerLite dialect)
//   CellTracker/track.py:11-114                        PR-GLS (legacy dialect)
//   CellTracker/tracker.py:1269-1289                   Gram + apply of one repetition
//
// Compiled with -ffp-contract=off: the fp64 geometry (distances, means, feature division) follows
// the reference's operation order with separate roundings so that the fp32 feature tables are
// bit-identical to numpy's; fma() is written explicitly where fusing is wanted.
//
// The (m*n) x 122 pair grid of the reference is never materialised: W2 acts linearly on the
// concatenation, so  concat(h_r, h_t) W2 = h_r W2[:512] + h_t W2[512:]  -- two small GEMMs plus an
// m x n x 512 element-wise/reduce kernel.
//
// The n x n system of the M-step,  (G diag(d) + c I)^T C^T = B^T  with G symmetric PSD, d >= 0,
// c = lambda sigma^2 > 0, is solved through the symmetric scaling
//   D^1/2 G D^1/2 + c I  =: M  (SPD),   x = D^1/2 M^-1 D^-1/2 b,
// i.e. a blocked Cholesky without pivoting (no D^-1 is formed: b_i / sqrt(d_i) is a weighted mean
// times sqrt(d_i), and rows with d_i = 0 decouple with x_i = 0 = b_i / c).
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <new>
#include <vector>

#include "../../include/ctamd.h"
#include "ct_exp.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

constexpr float kLeakyAlpha = 0.3f;
constexpr float kBnEps = 1e-3f;
constexpr int FEAT = 61, HID = 512;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Entry points that take a handle run on the handle's device whatever the calling thread's current device is
// (restored on return so that torch's view of the current device is not changed behind its back).
struct DeviceGuard {
    int prev = -1, want; hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) : want(device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) err = hipSetDevice(want);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

// Kernels whose working set lives in LDS need more than the 64 KiB default.  hipFuncSetAttribute is a per-device setting
// and the match paths run from several host threads (parallel.chain_map / FramePipeline): one bit per device, set after
// the (idempotent) call has succeeded, so a concurrent first use on the same device merely sets the attribute twice.
int ensure_big_lds(const void* fn, std::atomic<uint64_t>& done, int bytes = 150 * 1024) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return CT_OK;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
    return CT_OK;
}
// (a kernel with static LDS of its own asks for what it needs: static + dynamic must stay within the CU's 160 KiB)
#define ENSURE_LDS(kernel, bytes) do { static std::atomic<uint64_t> done_{0}; int rc_ = ensure_big_lds((const void*)kernel, done_, bytes); \
                                       if (rc_ != CT_OK) return rc_; } while (0)
#define ENSURE_BIG_LDS(kernel) do { static std::atomic<uint64_t> done_{0}; int rc_ = ensure_big_lds((const void*)kernel, done_); \
                                    if (rc_ != CT_OK) return rc_; } while (0)

// ------------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src); hi = __shfl(hi, src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    return v;
}


// ------------------------------------------------------------------------------------------------
// K6: kNN shape-context features (ffn.py:288-304)
// one wave per point; distances in LDS; k+1 rounds of wave arg-min on (distance, index)
// ------------------------------------------------------------------------------------------------
constexpr int KNN_MAXN = 4096;

// Batched EM (ct_prgls_two_ref_batched): problem b = blockIdx.z works on the same workspace layout shifted by b * stride
// doubles (inputs are copied into the workspace first, so EVERY pointer of a launch shifts alike); dims[4 b ..] = m, n, l of
// problem b (the grid is sized for the largest).  A single-problem launch has gridDim.z = 1: zero shift, dims == null.
struct Bt { size_t stride; const int* dims; };
#define BT_SHIFT(T, p) p = (T)((const double*)(p) + (size_t)blockIdx.z * bt.stride)
// (the legacy prediction chain batches its other kernels the same way: BT_DIM_N / BT_DIM_M read a problem's own sizes)
#define BT_DIM_M(v) do { if (bt.dims) v = bt.dims[4 * blockIdx.z + 0]; } while (0)
#define BT_DIM_N(v) do { if (bt.dims) v = bt.dims[4 * blockIdx.z + 1]; } while (0)
#define BT_DIM_L(v) do { if (bt.dims) v = bt.dims[4 * blockIdx.z + 2]; } while (0)

__global__ __launch_bounds__(64) void knn_features_kernel(const double* __restrict__ pts, int n, int k,
                                                          float* __restrict__ feat, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, pts); BT_SHIFT(float*, feat); BT_DIM_N(n);
    __shared__ double dist[KNN_MAXN];
    __shared__ double sel_d[32];
    __shared__ int sel_i[32];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    for (int j = lane; j < n; j += 64) {
        const double dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
        dist[j] = sqrt(dx * dx + dy * dy + dz * dz);       // separate roundings (no contraction)
    }
    __syncthreads();
    for (int round = 0; round <= k; ++round) {
        double best = INFINITY; int bi = 0x7fffffff;
        for (int j = lane; j < n; j += 64) {
            const double d = dist[j];
            if (d < best) { best = d; bi = j; }              // ascending j => lowest index on ties
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double od = shfl_xor_d(best, m); const int oi = __shfl_xor(bi, m);
            if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
        }
        if (lane == 0) { sel_d[round] = best; sel_i[round] = bi; dist[bi] = INFINITY; }
        __syncthreads();
    }
    // mean of the k+1 distances in numpy's pairwise-sum order (8 partial sums, then the tail)
    const int cnt = k + 1;
    double mean;
    {
        double res;
        if (cnt < 8) { res = 0.0; for (int q = 0; q < cnt; ++q) res += sel_d[q]; }
        else {
            double r[8];
            for (int q = 0; q < 8; ++q) r[q] = sel_d[q];
            int q = 8;
            for (; q < cnt - (cnt % 8); q += 8)
                for (int e = 0; e < 8; ++e) r[e] += sel_d[q + e];
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; q < cnt; ++q) res += sel_d[q];
        }
        mean = res / (double)cnt;
    }
    const int self = sel_i[0];
    const double sx = pts[3 * self], sy = pts[3 * self + 1], sz = pts[3 * self + 2];
    float* out = feat + (size_t)i * (3 * k + 1);
    for (int q = lane; q < k; q += 64) {
        const int nb = sel_i[q + 1];
        out[3 * q]     = (float)((pts[3 * nb] - sx) / mean);
        out[3 * q + 1] = (float)((pts[3 * nb + 1] - sy) / mean);
        out[3 * q + 2] = (float)((pts[3 * nb + 2] - sz) / mean);
    }
    if (lane == 0) out[3 * k] = (float)mean;
}

// ------------------------------------------------------------------------------------------------
// K14 normalize_points (ffn.py:330-374): centre; divide by 3 x std (ddof 0) of the projection on the first principal
// axis.  That std is sqrt(lambda_max(Xc^T Xc) / n), so no projection is needed: one workgroup computes the mean, the
// 3 x 3 scatter matrix (fp64, two-pass) and its largest eigenvalue by cyclic Jacobi rotations.
// para [dev] = mean[3], scale.  If `apply_para` is non-null the given (mean, scale) are used instead (normalising a
// second point set with the first one's parameters, trackerlite.py:91-93).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_points_kernel(const double* __restrict__ pts, int n, const double* __restrict__ apply_para,
                                                               double* __restrict__ out, double* __restrict__ para) {
    __shared__ double red[4][6];
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    if (!apply_para) {
        double s[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i < n; i += 256) { s[0] += pts[3 * i]; s[1] += pts[3 * i + 1]; s[2] += pts[3 * i + 2]; }
        for (int d = 0; d < 3; ++d) s[d] = wave_sum_d(s[d]);
        if ((tid & 63) == 0) for (int d = 0; d < 3; ++d) red[tid >> 6][d] = s[d];
        __syncthreads();
        if (tid == 0) for (int d = 0; d < 3; ++d) sh[d] = ((red[0][d] + red[1][d]) + (red[2][d] + red[3][d])) / (double)n;
        __syncthreads();
        const double mx = sh[0], my = sh[1], mz = sh[2];
        double c[6] = {0, 0, 0, 0, 0, 0};                  // xx xy xz yy yz zz
        for (int i = tid; i < n; i += 256) {
            const double x = pts[3 * i] - mx, y = pts[3 * i + 1] - my, z = pts[3 * i + 2] - mz;
            c[0] += x * x; c[1] += x * y; c[2] += x * z; c[3] += y * y; c[4] += y * z; c[5] += z * z;
        }
        for (int q = 0; q < 6; ++q) c[q] = wave_sum_d(c[q]);
        __syncthreads();
        if ((tid & 63) == 0) for (int q = 0; q < 6; ++q) red[tid >> 6][q] = c[q];
        __syncthreads();
        if (tid == 0) {
            double a[3][3];
            double t6[6];
            for (int q = 0; q < 6; ++q) t6[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
            a[0][0] = t6[0]; a[0][1] = a[1][0] = t6[1]; a[0][2] = a[2][0] = t6[2]; a[1][1] = t6[3]; a[1][2] = a[2][1] = t6[4]; a[2][2] = t6[5];
            for (int sweep = 0; sweep < 30; ++sweep) {
                const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
                if (off == 0.0) break;
                for (int p2 = 0; p2 < 2; ++p2)
                    for (int q2 = p2 + 1; q2 < 3; ++q2) {
                        if (a[p2][q2] == 0.0) continue;
                        const double theta = (a[q2][q2] - a[p2][p2]) / (2.0 * a[p2][q2]);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                        for (int k = 0; k < 3; ++k) { const double akp = a[k][p2], akq = a[k][q2]; a[k][p2] = cs * akp - sn * akq; a[k][q2] = sn * akp + cs * akq; }
                        for (int k = 0; k < 3; ++k) { const double apk = a[p2][k], aqk = a[q2][k]; a[p2][k] = cs * apk - sn * aqk; a[q2][k] = sn * apk + cs * aqk; }
                    }
            }
            const double lmax = fmax(a[0][0], fmax(a[1][1], a[2][2]));
            sh[3] = 3.0 * sqrt(lmax / (double)n);
            para[0] = sh[0]; para[1] = sh[1]; para[2] = sh[2]; para[3] = sh[3];
        }
        __syncthreads();
    } else {
        if (tid < 4) sh[tid] = apply_para[tid];
        __syncthreads();
    }
    if (out) {
        const double mx = sh[0], my = sh[1], mz = sh[2], sc = sh[3];
        for (int i = tid; i < n; i += 256) {
            out[3 * i] = (pts[3 * i] - mx) / sc; out[3 * i + 1] = (pts[3 * i + 1] - my) / sc; out[3 * i + 2] = (pts[3 * i + 2] - mz) / sc;
        }
    }
}

// x * scale + mean  (trackerlite.py:101)
__global__ void denormalize_points_kernel(const double* __restrict__ pts, int n, const double* __restrict__ para, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n) out[i] = pts[i] * para[3] + para[i % 3];
}

// ------------------------------------------------------------------------------------------------
// small fp32 GEMM  C[M][N] = A[M][K] (row stride lda) * B[K][N], optional BN-affine + LeakyReLU
// epilogue (Dense(no bias) + BatchNormalization + LeakyReLU, ffn.py:242-254).
// 64x64 tile, 256 threads, 4x4 outputs per thread, sequential-k fp32 accumulation.
// ------------------------------------------------------------------------------------------------
static bool gemm_valu() { static const bool v = getenv("CT_GEMM_VALU") && getenv("CT_GEMM_VALU")[0] == '1'; return v; }   // (A/B + the bit-identity test: the vector-fma form)
// MFMA = true (round 6, the default): the 64 x 64 block's inner products on the matrix cores' exact-fp32 instruction -- v_mfma_f32_16x16x4_f32 is an fmaf chain over its
// four k in ascending order, so every output still accumulates fmaf(a, b, acc) over k = 0, 1, 2, ...: the SAME bits as the vector form (tests/test_gpu_match.py compares the
// two), at a sixteenth of the vector-issue slots (one MFMA per 2048 flop instead of 16 v_fma: the match stream runs beside the U-Net's issue-bound thin layers) and twice the
// rate of un-packed v_fma.  A wave takes a 32 x 32 quarter of the block as 2 x 2 MFMA tiles; the panels, their staging and the epilogue arithmetic are unchanged.
template <bool MFMA>
__device__ __forceinline__ void gemm_f32_body(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                              float* __restrict__ C, int M, int N, int K,
                                              const float* __restrict__ bn /* [4][N] gamma,beta,mean,var or null */) {
    if ((int)blockIdx.y * 64 >= M) return;
    // K in steps of GK = 32 through two LDS buffers: the next step's 64 x 32 | 32 x 64 panels are fetched into registers before the current
    // step's FMAs and parked after them, one barrier per step (the first version waited a full L2 round trip per 16-deep step with nothing
    // else in flight: 60 us per 600 x 512 x 512 product, four of them per match).  Every output still accumulates fmaf(a, b, acc) over k = 0,
    // 1, 2, ... : the same bits.
    constexpr int GK = 32;
    __shared__ float As[2][GK][64 + 1];
    __shared__ float Bs[2][GK][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    typedef float gf32x4 __attribute__((ext_vector_type(4)));
    gf32x4 macc[2][2] = {{gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}}, {gf32x4{0.f, 0.f, 0.f, 0.f}, gf32x4{0.f, 0.f, 0.f, 0.f}}};
    const int lane = tid & 63, wv = tid >> 6, wr = 32 * (wv >> 1), wc = 32 * (wv & 1), l16 = lane & 15, lk = lane >> 4;
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            const int r = e >> 5, kk = e & 31;                       // A panel: 64 rows x 32 k
            const int gm = m0 + r, gk = k0 + kk;
            ra[u] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.f;
            const int kb = e >> 6, c = e & 63;                       // B panel: 32 k x 64 columns
            const int gkb = k0 + kb, gn = n0 + c;
            rb[u] = (gkb < K && gn < N) ? B[(size_t)gkb * N + gn] : 0.f;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u;
            As[buf][e & 31][e >> 5] = ra[u];
            Bs[buf][e >> 6][e & 63] = rb[u];
        }
    };
    fetch(0); park(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += GK) {
        const bool more = k0 + GK < K;
        if (more) fetch(k0 + GK);
        if constexpr (MFMA) {
#pragma unroll
            for (int k4 = 0; k4 < GK; k4 += 4) {                  // lane (l16, lk): A[row l16][k4 + lk], B[k4 + lk][col l16]; D rows 4 lk + e
                const float a0 = As[buf][k4 + lk][wr + l16], a1 = As[buf][k4 + lk][wr + 16 + l16];
                const float b0 = Bs[buf][k4 + lk][wc + l16], b1 = Bs[buf][k4 + lk][wc + 16 + l16];
                macc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, macc[0][0], 0, 0, 0);
                macc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, macc[0][1], 0, 0, 0);
                macc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, macc[1][0], 0, 0, 0);
                macc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, macc[1][1], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[buf][kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        }
        if (more) park(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // the lane's 16 outputs: vector form rows ty * 4 + i, columns tx * 4 + j; MFMA form tile (i >> 1, j >> 1) ... written as one loop over (i, j) in 0..3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = MFMA ? m0 + wr + 16 * (i >> 1) + 4 * lk + 2 * (i & 1) : m0 + ty * 4 + i;     // MFMA: (i, j) -> tile i >> 1, element e = 2 (i & 1) + (j & 1), column tile j >> 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 2 * (i & 1) + (j & 1);
            const int gmm = MFMA ? m0 + wr + 16 * (i >> 1) + 4 * lk + e : gm;
            const int gn = MFMA ? n0 + wc + 16 * (j >> 1) + l16 : n0 + tx * 4 + j;
            if (gmm >= M || gn >= N) continue;
            float v = MFMA ? macc[i >> 1][j >> 1][e] : acc[i][j];
            if (bn) {
                const float inv = bn[gn] / sqrtf(bn[3 * N + gn] + kBnEps);
                v = (v - bn[2 * N + gn]) * inv + bn[N + gn];
                v = v >= 0.f ? v : v * kLeakyAlpha;
            }
            C[(size_t)gmm * N + gn] = v;
        }
    }
}
template <bool MFMA>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K,
                                                       const float* __restrict__ bn /* [4][N] gamma,beta,mean,var or null */,
                                                       Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const float*, A); BT_SHIFT(float*, C); BT_DIM_N(M);         // batched: rows = the problem's reference points
    gemm_f32_body<MFMA>(A, lda, B, C, M, N, K, bn);
}
// two independent products of one shape family in ONE launch (blockIdx.z picks the operands): the FFN's reference-side and target-side
// layers.  Beside the U-Net every launch of the match stream waits for workgroup slots (a 600 x 512 x 512 product: 27 us alone, 214 us in
// the frame loop); the products themselves are unchanged.
struct GemmPair { const float* A[2]; const float* B[2]; float* C[2]; int M[2]; };
template <bool MFMA>
__global__ __launch_bounds__(256) void gemm_f32_pair_kernel(GemmPair g, int lda, int N, int K, const float* __restrict__ bn) {
    const int z = blockIdx.z;
    gemm_f32_body<MFMA>(g.A[z], lda, g.B[z], g.C[z], g.M[z], N, K, bn);
}

// ------------------------------------------------------------------------------------------------
// K7 pair kernel: corr[t][r] = sigmoid( w3 . leaky(BN2(U[r] + V[t])) + b3 )
// 32 x 32 pairs per block, 256 threads, 2 x 2 pairs per thread, k streamed in chunks of 64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ffn_pair_kernel(const float* __restrict__ U, int n, const float* __restrict__ V, int m,
                                                       const float* __restrict__ bn2, const float* __restrict__ w3,
                                                       float b3, float* __restrict__ corr, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const float*, U); BT_SHIFT(float*, corr); BT_DIM_N(n);      // V: shared by the batch (legacy chain) unless dims[3] != 0
    if (bt.dims && bt.dims[4 * blockIdx.z + 3]) { BT_SHIFT(const float*, V); BT_DIM_M(m); }
    if ((int)blockIdx.x * 32 >= n || (int)blockIdx.y * 32 >= m) return;
    constexpr int KC = 64;
    __shared__ float Us[32][KC + 1];
    __shared__ float Vs[32][KC + 1];
    __shared__ float inv_s[KC], mean_s[KC], beta_s[KC], w3_s[KC];
    const int tid = threadIdx.x, tr = tid & 15, tt = tid >> 4;
    const int r0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    float acc[2][2] = {};
    for (int k0 = 0; k0 < HID; k0 += KC) {
        for (int e = tid; e < 32 * KC; e += 256) {
            const int row = e / KC, kk = e - row * KC;
            Us[row][kk] = (r0 + row < n) ? U[(size_t)(r0 + row) * HID + k0 + kk] : 0.f;
            Vs[row][kk] = (t0 + row < m) ? V[(size_t)(t0 + row) * HID + k0 + kk] : 0.f;
        }
        if (tid < KC) {
            const int kk = k0 + tid;
            inv_s[tid] = bn2[kk] / sqrtf(bn2[3 * HID + kk] + kBnEps);
            mean_s[tid] = bn2[2 * HID + kk]; beta_s[tid] = bn2[HID + kk]; w3_s[tid] = w3[kk];
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < KC; ++kk) {
            const float u0 = Us[tr][kk], u1 = Us[tr + 16][kk];
            const float v0 = Vs[tt][kk], v1 = Vs[tt + 16][kk];
            const float inv = inv_s[kk], mu = mean_s[kk], be = beta_s[kk], ww = w3_s[kk];
            float y;
            y = ((u0 + v0) - mu) * inv + be; y = y >= 0.f ? y : y * kLeakyAlpha; acc[0][0] = fmaf(y, ww, acc[0][0]);
            y = ((u1 + v0) - mu) * inv + be; y = y >= 0.f ? y : y * kLeakyAlpha; acc[0][1] = fmaf(y, ww, acc[0][1]);
            y = ((u0 + v1) - mu) * inv + be; y = y >= 0.f ? y : y * kLeakyAlpha; acc[1][0] = fmaf(y, ww, acc[1][0]);
            y = ((u1 + v1) - mu) * inv + be; y = y >= 0.f ? y : y * kLeakyAlpha; acc[1][1] = fmaf(y, ww, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + tt + 16 * i, r = r0 + tr + 16 * j;
            if (t < m && r < n) corr[(size_t)t * n + r] = 1.f / (1.f + expf(-(acc[i][j] + b3)));
        }
}

// rows of explicit 122-feature inputs: out[i] = sigmoid(w3 . leaky(BN2(U[i] + V[i])) + b3)
__global__ __launch_bounds__(64) void ffn_rows_finish_kernel(const float* __restrict__ U, const float* __restrict__ V, int rows,
                                                             const float* __restrict__ bn2, const float* __restrict__ w3,
                                                             float b3, float* __restrict__ out) {
    const int i = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int k = lane; k < HID; k += 64) {
        const float inv = bn2[k] / sqrtf(bn2[3 * HID + k] + kBnEps);
        float y = ((U[(size_t)i * HID + k] + V[(size_t)i * HID + k]) - bn2[2 * HID + k]) * inv + bn2[HID + k];
        y = y >= 0.f ? y : y * kLeakyAlpha;
        acc = fmaf(y, w3[k], acc);
    }
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) acc += __shfl_xor(acc, mk);
    if (lane == 0) out[i] = 1.f / (1.f + expf(-(acc + b3)));
}

// ------------------------------------------------------------------------------------------------
// K8 greedy one-to-one assignment (trackerlite.py:242-259 / track.py:58-70)
// Single workgroup (the loop is inherently sequential): per-row best (value, column) over free
// columns is cached in LDS; every step = block arg-max over rows (ties -> lowest row, whose cached
// column is the lowest column) then a re-scan of only those rows whose cached column was taken.
// Equivalent to the reference's repeated global arg-max with first-occurrence tie-breaking.
// ------------------------------------------------------------------------------------------------
// The reference repeats "global arg-max (first occurrence on ties) -> record -> clear its row and column" up to n
// times: inherently sequential.  With the strict total order  e1 > e2  <=>  (value1 > value2) or (equal values and
// flat index1 < flat index2)  the greedy matching is exactly the set obtained by repeatedly accepting every
// LOCALLY DOMINANT edge (best of its row AND best of its column among the still-free rows/columns): the current
// global maximum is always locally dominant, and accepting another dominant edge first never changes what the
// sequential loop would pick.  Each round is three data-parallel kernels; the number of rounds is O(log n) for
// generic inputs.  The threshold test of the reference (stop at the first maximum below it) == never accept an
// edge below it.  A final sort by the same order recovers the reference's pick sequence.
constexpr int GD_ROWLANES = 16;      // single match: 1024-thread workgroups keep the column pass short
// batched chains share the chip with other streams: a 1024-thread workgroup needs 4 free wave slots on EVERY SIMD of one CU and
// starves behind the 256-thread workgroups of the neighbouring chains (466 us per launch under the frame pipeline against 10 us
// alone), so they use 256 threads.  Both reductions pick (max value, lowest index): the result does not depend on the split.
constexpr int GD_ROWLANES_BATCH = 4;
constexpr int GD_UN = 8;                        // loop steps whose loads are in flight together
enum { GD_COUNT = 0, GD_NEW = 1 /* and 2: one counter per round parity */, GD_DONE = 3 };

// One round = two launches.  gd_best_kernel: blocks [0, nrb) find every free row's best free column (16 rows per block, one wave
// each), blocks [nrb, nrb + ncb) every free column's best free row (64 columns x 16 row lanes); gd_accept_kernel accepts the
// locally dominant edges.  Termination without a separate launch: accept of round k counts into NEW[k & 1]; round k + 1 starts by
// looking at that counter (zero -> nothing was accepted -> done, every block leaves) and re-arms NEW[(k + 1) & 1], which nobody
// touches in between.  (Round 1: four launches per round -- row best, column best, accept, round end.)
template <int RL>
__global__ __launch_bounds__(64 * RL) void gd_best_kernel(const float* __restrict__ corr, int m, int n,
                                                                   const unsigned char* __restrict__ row_used,
                                                                   const unsigned char* __restrict__ col_used,
                                                                   float* __restrict__ rowval, int* __restrict__ rowcol,
                                                                   int* __restrict__ colrow, int* __restrict__ ctr, int round, int nrb,
                                                                   Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const float*, corr); BT_SHIFT(const unsigned char*, row_used); BT_SHIFT(const unsigned char*, col_used);
    BT_SHIFT(float*, rowval); BT_SHIFT(int*, rowcol); BT_SHIFT(int*, colrow); BT_SHIFT(int*, ctr); BT_DIM_N(n); BT_DIM_M(m);
    if (ctr[GD_DONE]) return;
    if (round > 0 && ctr[GD_NEW + ((round - 1) & 1)] == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctr[GD_DONE] = 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr[GD_NEW + (round & 1)] = 0;      // re-arm this round's counter
    if ((int)blockIdx.x < nrb) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int t = blockIdx.x * RL + wave;
        if (t >= m) return;
        if (row_used[t]) { if (lane == 0) rowcol[t] = -1; return; }
        const float* row = corr + (size_t)t * n;
        float best = -1.f; int bi = 0x7fffffff;
        // GD_UN steps' loads are issued before the first compare (a `continue` in front of the load makes every step wait for the
        // one before it: a chain of dependent round trips that stretched 20x beside other streams); same compares, same order
        for (int c0 = lane; c0 < n; c0 += 64 * GD_UN) {
            float vv[GD_UN]; unsigned char uu[GD_UN];
#pragma unroll
            for (int u = 0; u < GD_UN; ++u) { const int c = min(c0 + 64 * u, n - 1); uu[u] = col_used[c]; vv[u] = row[c]; }
#pragma unroll
            for (int u = 0; u < GD_UN; ++u) {
                const int c = c0 + 64 * u;
                if (c < n && !uu[u] && vv[u] > best) { best = vv[u]; bi = c; }   // ascending c => lowest column on ties
            }
        }
#pragma unroll
        for (int mk = 32; mk >= 1; mk >>= 1) {
            const float ov = __shfl_xor(best, mk); const int oi = __shfl_xor(bi, mk);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { rowval[t] = best; rowcol[t] = (bi == 0x7fffffff) ? -1 : bi; }
        return;
    }
    __shared__ float sv[RL][64];
    __shared__ int si[RL][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int r = ((int)blockIdx.x - nrb) * 64 + cl;
    float best = -1.f; int btg = 0x7fffffff;
    if (r < n && !col_used[r])
        for (int t0 = rl; t0 < m; t0 += RL * GD_UN) {
            float vv[GD_UN]; unsigned char uu[GD_UN];
#pragma unroll
            for (int u = 0; u < GD_UN; ++u) { const int t = min(t0 + RL * u, m - 1); uu[u] = row_used[t]; vv[u] = corr[(size_t)t * n + r]; }
#pragma unroll
            for (int u = 0; u < GD_UN; ++u) {
                const int t = t0 + RL * u;
                if (t < m && !uu[u] && vv[u] > best) { best = vv[u]; btg = t; }  // ascending t => lowest row on ties
            }
        }
    sv[rl][cl] = best; si[rl][cl] = btg;
    __syncthreads();
    if (rl == 0 && r < n) {
        for (int q = 1; q < RL; ++q) {
            const float ov = sv[q][cl]; const int oi = si[q][cl];
            if (ov > best || (ov == best && oi < btg)) { best = ov; btg = oi; }
        }
        colrow[r] = (btg == 0x7fffffff) ? -1 : btg;
    }
}

__global__ __launch_bounds__(256) void gd_accept_kernel(int m, int n, float thr, const float* __restrict__ rowval,
                                                        const int* __restrict__ rowcol, const int* __restrict__ colrow,
                                                        unsigned char* __restrict__ row_used, unsigned char* __restrict__ col_used,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ ctr, int round,
                                                        Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const float*, rowval); BT_SHIFT(const int*, rowcol); BT_SHIFT(const int*, colrow); BT_SHIFT(unsigned char*, row_used);
    BT_SHIFT(unsigned char*, col_used); BT_SHIFT(unsigned long long*, keys); BT_SHIFT(int*, ctr); BT_DIM_N(n); BT_DIM_M(m);
    if (ctr[GD_DONE]) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= m || row_used[t]) return;
    const int r = rowcol[t];
    const float v = rowval[t];
    if (r < 0 || !(v >= thr) || colrow[r] != t) return;
    row_used[t] = 1; col_used[r] = 1;
    const int k = atomicAdd(&ctr[GD_COUNT], 1);
    atomicAdd(&ctr[GD_NEW + (round & 1)], 1);
    const unsigned flat = (unsigned)t * (unsigned)n + (unsigned)r;
    keys[k] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - flat);
}

// ------------------------------------------------------------------------------------------------
// All rounds of the greedy assignment in ONE launch (the single-match path, ct_greedy_match): the loop "row / column best -> accept -> anything
// new?" is sequential anyway, and as 11-19 pairs of launches each of its ~30 tiny kernels waited for a workgroup slot beside the U-Net (the frame
// loop: gd_best 75-130 us per launch against 8 us alone, ~1 ms of a frame's match stream).  A small persistent grid walks the rounds with a
// device-side barrier between the two phases; per round exactly gd_best_kernel's and gd_accept_kernel's arithmetic on the same order
// (max value, lowest index), so the accepted set and the pick order are the same bit for bit.
//   * the grid is small enough to be co-resident several times over on the CUs the launch stream may use (the host sizes it from the
//     stream's CU mask: a barrier between workgroups that cannot all be resident would never open);
//   * what a round writes and the next reads travels through L2: the barrier is a release (__threadfence in EVERY wave before the arrival) and an
//     acquire (after the last arrival), and the wave-uniform reads (flags of a wave's own row, the counters) are agent-scope loads -- plain
//     loads become scalar loads, and the scalar cache is not covered by the vector L1's invalidate (ct_fresh.h has the history).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gd_fresh_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned char gd_fresh_u8(const unsigned char* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// How many workgroups of `kernel` the runtime will keep resident per CU (registers, LDS and wave slots considered); 1 if it cannot say.
template <class K> static int coresident_per_cu(K kernel, int threads, size_t dyn_lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, dyn_lds) != hipSuccess) { (void)hipGetLastError(); nb = 1; }
    return nb < 1 ? 1 : nb;
}
enum { GD_BAR = 8 };                          // ctr[GD_BAR]: arrivals at the device-side barrier (monotonic)
__device__ __forceinline__ void gd_grid_barrier(int* bar, int nblk, int& passed) {
    __threadfence();                                                              // EVERY wave releases its own stores / no-return atomics of the phase: the
    __syncthreads();                                                              // workgroup barrier alone does not wait for vmcnt (round-5 advisor finding)
    ++passed;
    if (threadIdx.x == 0) {
        __threadfence();                                                          // release: this workgroup's writes of the phase
        atomicAdd(bar, 1);
        while (gd_fresh_i32(bar) < passed * nblk) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                                          // acquire: the other workgroups' writes
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void gd_persistent_kernel(const float* __restrict__ corr, int m, int n, float thr, unsigned char* row_used,
                                                            unsigned char* col_used, float* rowval, int* rowcol, int* colrow,
                                                            unsigned long long* keys, int* ctr, int max_rounds) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int G = gridDim.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ncb = (n + 63) / 64;
    int passed = 0;
    for (int round = 0; round < max_rounds; ++round) {
        if (round > 0 && gd_fresh_i32(&ctr[GD_NEW + ((round - 1) & 1)]) == 0) break;     // the last round accepted nothing (every workgroup reads the same word)
        if (blockIdx.x == 0 && threadIdx.x == 0) ctr[GD_NEW + (round & 1)] = 0;            // re-arm this round's counter (last read two barriers ago)
        // ---- every free row's best free column (a wave per row; gd_best_kernel's row blocks)
        for (int t = blockIdx.x * 4 + wave; t < m; t += G * 4) {
            if (gd_fresh_u8(&row_used[t])) { if (lane == 0) rowcol[t] = -1; continue; }
            const float* row = corr + (size_t)t * n;
            float best = -1.f; int bi = 0x7fffffff;
            for (int c0 = lane; c0 < n; c0 += 64 * GD_UN) {
                float vv[GD_UN]; unsigned char uu[GD_UN];
#pragma unroll
                for (int u = 0; u < GD_UN; ++u) { const int c = min(c0 + 64 * u, n - 1); uu[u] = col_used[c]; vv[u] = row[c]; }
#pragma unroll
                for (int u = 0; u < GD_UN; ++u) {
                    const int c = c0 + 64 * u;
                    if (c < n && !uu[u] && vv[u] > best) { best = vv[u]; bi = c; }   // ascending c => lowest column on ties
                }
            }
#pragma unroll
            for (int mk = 32; mk >= 1; mk >>= 1) {
                const float ov = __shfl_xor(best, mk); const int oi = __shfl_xor(bi, mk);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { rowval[t] = best; rowcol[t] = (bi == 0x7fffffff) ? -1 : bi; }
        }
        // ---- every free column's best free row (64 columns x 4 row lanes per workgroup pass; gd_best_kernel's column blocks)
        for (int cb = blockIdx.x; cb < ncb; cb += G) {
            const int r = cb * 64 + lane;
            float best = -1.f; int btg = 0x7fffffff;
            if (r < n && !col_used[r])
                for (int t0 = wave; t0 < m; t0 += 4 * GD_UN) {
                    float vv[GD_UN]; unsigned char uu[GD_UN];
#pragma unroll
                    for (int u = 0; u < GD_UN; ++u) { const int t = min(t0 + 4 * u, m - 1); uu[u] = gd_fresh_u8(&row_used[t]); vv[u] = corr[(size_t)t * n + r]; }
#pragma unroll
                    for (int u = 0; u < GD_UN; ++u) {
                        const int t = t0 + 4 * u;
                        if (t < m && !uu[u] && vv[u] > best) { best = vv[u]; btg = t; }  // ascending t => lowest row on ties
                    }
                }
            sv[wave][lane] = best; si[wave][lane] = btg;
            __syncthreads();
            if (wave == 0 && r < n) {
                for (int q = 1; q < 4; ++q) {
                    const float ov = sv[q][lane]; const int oi = si[q][lane];
                    if (ov > best || (ov == best && oi < btg)) { best = ov; btg = oi; }
                }
                colrow[r] = (btg == 0x7fffffff) ? -1 : btg;
            }
            __syncthreads();
        }
        gd_grid_barrier(&ctr[GD_BAR], G, passed);
        // ---- accept the locally dominant edges (gd_accept_kernel)
        for (int t = blockIdx.x * 256 + threadIdx.x; t < m; t += G * 256) {
            if (row_used[t]) continue;
            const int r = rowcol[t];
            const float v = rowval[t];
            if (r < 0 || !(v >= thr) || colrow[r] != t) continue;
            row_used[t] = 1; col_used[r] = 1;
            const int k = atomicAdd(&ctr[GD_COUNT], 1);
            atomicAdd(&ctr[GD_NEW + (round & 1)], 1);
            const unsigned flat = (unsigned)t * (unsigned)n + (unsigned)r;
            keys[k] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - flat);
        }
        gd_grid_barrier(&ctr[GD_BAR], G, passed);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr[GD_DONE] = 1;
}

// single workgroup: bitonic sort of the accepted keys (descending) -> pairs in the reference's pick order
__global__ __launch_bounds__(1024) void gd_finalize_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ ctr,
                                                           int n, int32_t* __restrict__ pairs, int32_t* __restrict__ n_pairs,
                                                           Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const unsigned long long*, keys); BT_SHIFT(const int*, ctr); BT_SHIFT(int32_t*, pairs); BT_SHIFT(int32_t*, n_pairs); BT_DIM_N(n);
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
    const int cnt = ctr[GD_COUNT];
    int p2 = 1; while (p2 < cnt) p2 <<= 1;
    for (int i = threadIdx.x; i < p2; i += 1024) sk[i] = i < cnt ? keys[i] : 0ull;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < p2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool desc = (i & k) == 0;
                    const unsigned long long x = sk[i], y = sk[ixj];
                    if (desc ? (x < y) : (x > y)) { sk[i] = y; sk[ixj] = x; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < cnt; i += 1024) {
        const unsigned flat = 0xFFFFFFFFu - (unsigned)(sk[i] & 0xFFFFFFFFull);
        pairs[2 * i] = (int)(flat % (unsigned)n); pairs[2 * i + 1] = (int)(flat / (unsigned)n);      // (ref, tgt)
    }
    if (threadIdx.x == 0) *n_pairs = cnt;
}

__global__ __launch_bounds__(256) void prior_fill_kernel(double* __restrict__ prior, int m, int n, int mode,
                                                         const int32_t* __restrict__ pairs, const int32_t* __restrict__ n_pairs,
                                                         int* __restrict__ row_match /* [m] scratch */, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(double*, prior); BT_SHIFT(const int32_t*, pairs); BT_SHIFT(const int32_t*, n_pairs); BT_SHIFT(int*, row_match); BT_DIM_N(n); BT_DIM_M(m);
    // pass A (blockIdx.y == 0): row_match[t] = matched ref or -1
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)m * n;
    if (gid >= tot) return;
    const int t = (int)(gid / n), r = (int)(gid - (size_t)t * n);
    const int mr = row_match[t];
    double v;
    if (mode == 0) {          // np.full_like(float32 matrix, 0.1/(n-1)); prior[tgt, ref] = 0.9   (float32 values)
        v = (double)(float)(0.1 / (double)(n - 1));
        if (mr == r) v = (double)0.9f;
    } else {                  // legacy: ones/n; matched rows 0.1/(n-1) with 0.9 at the pair        (float64 values)
        v = 1.0 / (double)n;
        if (mr >= 0) v = (mr == r) ? 0.9 : 0.1 / (double)(n - 1);
    }
    prior[gid] = v;
}

__global__ void row_match_kernel(int* __restrict__ row_match, int m, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(int*, row_match); BT_DIM_M(m);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < m) row_match[tid] = -1;
}
__global__ void row_match_set_kernel(int* __restrict__ row_match, const int32_t* __restrict__ pairs,
                                     const int32_t* __restrict__ n_pairs, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(int*, row_match); BT_SHIFT(const int32_t*, pairs); BT_SHIFT(const int32_t*, n_pairs);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < *n_pairs) row_match[pairs[2 * tid + 1]] = pairs[2 * tid];
}

// ------------------------------------------------------------------------------------------------
// PR-GLS kernels (fp64)
// ------------------------------------------------------------------------------------------------
// scalars block (device): [0] sigma2  [1] gamma  [2] sumP  [3] move_norm2  [4] c = lambda*sigma2  [5] iteration
enum { S_SIGMA2 = 0, S_GAMMA = 1, S_SUMP = 2, S_NORM2 = 3, S_C = 4, S_IT = 5, S_DONE = 6, S_RES = 7, S_NUM = 8 };

// Low-rank M-step (see "Low-rank fast path" below): tolerances of the nested pivoted-Cholesky factorisation |G - U^T U|_max
// (diag(G) = 1) and the relative residuals of the exact system at which the rank is raised / the step is rejected.
constexpr double kLowRankTol = 1e-10;              // coarse rank (~40): enough while c = lambda sigma2 is large
constexpr double kLowRankTolTight = 1e-13;         // fine rank (~65): needed once sigma2 has shrunk (c small, truncation / c matters)
constexpr double kLowRankSwitchResidual = 2e-7;    // device-side: the residual predicted for the next iteration above this -> fine rank from then on
constexpr double kLowRankMaxResidual = 1e-6;       // host-side: the chunk is redone (fine rank, then dense)

// out[i][j] = exp(-|a_j - b_i|^2 / (2 s2)),  i < nb, j < na      (trackerlite.py:368-372)
__global__ __launch_bounds__(256) void gauss_kernel(const double* __restrict__ a, int na, const double* __restrict__ b, int nb,
                                                    double two_s2, double* __restrict__ out, int raw = 0, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, a); BT_SHIFT(const double*, b); BT_SHIFT(double*, out);
    if (bt.dims) { na = nb = bt.dims[4 * blockIdx.z + 1]; }               // batched use: the Gram matrix of a problem's own reference set
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)na * nb) return;
    const int i = (int)(gid / na), j = (int)(gid - (size_t)i * na);
    const double dx = a[3 * j] - b[3 * i], dy = a[3 * j + 1] - b[3 * i + 1], dz = a[3 * j + 2] - b[3 * i + 2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    out[gid] = raw ? d2 : exp(-d2 / two_s2);
}

// sum over all (t, r) of |ref_r - tgt_t|^2  -> per-row partial sums rowpart[m]
__global__ __launch_bounds__(256) void dist2_rowsum_kernel(const double* __restrict__ ref, int n, const double* __restrict__ tgt, int m,
                                                           const double* __restrict__ P /* weights or null */,
                                                           double* __restrict__ rowpart, const double* __restrict__ sc = nullptr,
                                                           Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, ref); BT_SHIFT(const double*, tgt); BT_SHIFT(double*, rowpart);
    if (P) BT_SHIFT(const double*, P);
    if (sc) BT_SHIFT(const double*, sc);
    if (bt.dims) { m = bt.dims[4 * blockIdx.z]; n = bt.dims[4 * blockIdx.z + 1]; }
    if (sc && sc[S_DONE] != 0.0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= m) return;
    const double yx = tgt[3 * t], yy = tgt[3 * t + 1], yz = tgt[3 * t + 2];
    double acc = 0.0;
    for (int r = lane; r < n; r += 64) {
        const double dx = ref[3 * r] - yx, dy = ref[3 * r + 1] - yy, dz = ref[3 * r + 2] - yz;
        const double d2 = dx * dx + dy * dy + dz * dz;
        acc += P ? d2 * P[(size_t)t * n + r] : d2;
    }
    acc = wave_sum_d(acc);
    if (lane == 0) rowpart[t] = acc;
}

// finalise scalars from row partials.  mode 0: initial sigma2 = sum / (3 m n)   (both dialects)
// mode 1 (lite, trackerlite.py:342-350):  gamma = max(1 - sumP/m, 1e-4); sigma2 = sum / (3 sumP)
// mode 2 (legacy, track.py:103-112):      gamma = 1 - sumP/m;            sigma2 = max(sum / (3 sumP), 1)
// em_persistent_kernel calls the bodies of the EM kernels for a VIRTUAL block with the iteration's scalars in registers: inside one launch
// the scalar block changes between phases, and a wave-uniform plain load of it may come back stale from the scalar cache (ct_fresh.h).
// ov == nullptr (every other caller): blockIdx / gridDim and the scalar block in memory, as before.
struct EmOv { int vb, nvb; double s2, gamma, c; int r; bool add; };
__device__ __forceinline__ double em_fresh_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int em_fresh_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (body of scalars_kernel; also the tail of apply_dual_scalars_kernel, where the last workgroup of the field application runs it)
template <bool FRESH = false>
__device__ __forceinline__ void scalars_body(const double* rowpart, int m, int n, int mode, double* sc, const double* normpart,
                                             const double* respart, int* rank_p, Bt bt, const double* tr_tgt,
                                             const double* tr_arow, const double* tr_pred, const double* tr_d, const double* tr_b) {
    auto SC = [&](int k) { return FRESH ? em_fresh_f64(sc + k) : sc[k]; };
    auto RK = [&](int k) { return FRESH ? em_fresh_i32(rank_p + k) : rank_p[k]; };
    __shared__ double red[4];
    __shared__ double red2[4];
    __shared__ double red3[4], red4[4];
    BT_SHIFT(const double*, rowpart); BT_SHIFT(double*, sc);
    if (normpart) BT_SHIFT(const double*, normpart);
    if (respart) BT_SHIFT(const double*, respart);
    if (rank_p) BT_SHIFT(int*, rank_p);
    if (bt.dims) { m = bt.dims[4 * blockIdx.z]; n = bt.dims[4 * blockIdx.z + 1]; }
    if (mode != 0 && SC(S_DONE) != 0.0) return;
    if (respart) {
        double r1 = 0.0, r2 = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) { r1 = fmax(r1, respart[i]); r2 = fmax(r2, respart[n + i]); }
#pragma unroll
        for (int mk = 32; mk >= 1; mk >>= 1) { r1 = fmax(r1, shfl_xor_d(r1, mk)); r2 = fmax(r2, shfl_xor_d(r2, mk)); }
        if ((threadIdx.x & 63) == 0) { red3[threadIdx.x >> 6] = r1; red4[threadIdx.x >> 6] = r2; }
    }
    double nacc = 0.0;
    if (normpart) {
        for (int i = threadIdx.x; i < n; i += 256) nacc += normpart[i];
        nacc = wave_sum_d(nacc);
        if ((threadIdx.x & 63) == 0) red2[threadIdx.x >> 6] = nacc;
    }
    double acc = 0.0;
    if (tr_arow) {
        // sum_tr P_tr |x_r - y_t|^2 = sum_t a_t |y_t|^2 - 2 sum_r x_r . b_r + sum_r d_r |x_r|^2   (a = row sums of P, d = column sums,
        // b = P^T Y: all by-products of the E-step; x = the reference set AFTER this iteration's movement).  O(m + n) instead of one
        // more pass over the m x n posterior; the cancellation (terms ~ sumP, result ~ 3 sigma2 sumP) costs ~1e-12 relative in sigma2
        // where the direct sum has ~1e-15 - far inside what the low-rank M-step already tolerates.
        BT_SHIFT(const double*, tr_tgt); BT_SHIFT(const double*, tr_arow); BT_SHIFT(const double*, tr_pred); BT_SHIFT(const double*, tr_d);
        BT_SHIFT(const double*, tr_b);
        for (int t = threadIdx.x; t < m; t += 256) {
            const double yx = tr_tgt[3 * t], yy = tr_tgt[3 * t + 1], yz = tr_tgt[3 * t + 2];
            acc += tr_arow[t] * (yx * yx + yy * yy + yz * yz);
        }
        for (int r = threadIdx.x; r < n; r += 256) {
            const double xx = tr_pred[3 * r], xy = tr_pred[3 * r + 1], xz = tr_pred[3 * r + 2];
            acc += tr_d[r] * (xx * xx + xy * xy + xz * xz) - 2.0 * (xx * tr_b[3 * r] + xy * tr_b[3 * r + 1] + xz * tr_b[3 * r + 2]);
        }
    } else
        for (int t = threadIdx.x; t < m; t += 256) acc += rowpart[t];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = (red[0] + red[1]) + (red[2] + red[3]);
        if (mode == 0) { sc[S_SIGMA2] = s / (3.0 * (double)m * (double)n); }
        else {
            const double sp = SC(S_SUMP);
            const double s2_prev = SC(S_SIGMA2);              // the sigma2 this iteration's M-step (and its residual) used
            double g = 1.0 - sp / (double)m;
            double s2 = s / (3.0 * sp);
            if (mode == 1) { if (g < 1e-4) g = 1e-4; }
            else { if (s2 < 1.0) s2 = 1.0; }
            sc[S_GAMMA] = g; sc[S_SIGMA2] = s2;
            sc[S_IT] = SC(S_IT) + 1.0;
            if (normpart) {
                const double n2 = (red2[0] + red2[1]) + (red2[2] + red2[3]);
                sc[S_NORM2] = n2;
                if (mode == 1 && sqrt(n2) < 1e-3) sc[S_DONE] = 1.0;           // trackerlite.py:353-356
            }
            if (respart) {
                const double r1 = fmax(fmax(red3[0], red3[1]), fmax(red3[2], red3[3]));
                const double r2 = fmax(fmax(red4[0], red4[1]), fmax(red4[2], red4[3]));
                const double rel = r2 > 0.0 ? r1 / r2 : (r1 > 0.0 ? INFINITY : 0.0);
                if (!(rel <= SC(S_RES))) sc[S_RES] = rel;
                // the monitor also steers the rank: well before the residual reaches the rejection level the following
                // iterations use the finer rows of the (nested) factorisation -- rank_p[0] current, rank_p[1] finest
                // (the truncation error of the M-step scales with 1 / c = 1 / (lambda sigma2): predict the next iteration's)
                const double predicted = s2 > 0.0 ? rel * (s2_prev / s2) : INFINITY;
                if (rank_p && !(fmax(rel, predicted) <= kLowRankSwitchResidual) && RK(0) < RK(1)) rank_p[0] = RK(1);
            }
        }
    }
}
__global__ __launch_bounds__(256) void scalars_kernel(const double* __restrict__ rowpart, int m, int n, int mode,
                                                      double* __restrict__ sc, const double* __restrict__ normpart = nullptr,
                                                      const double* __restrict__ respart = nullptr, int* __restrict__ rank_p = nullptr,
                                                      Bt bt = Bt{0, nullptr}, const double* __restrict__ tr_tgt = nullptr,
                                                      const double* __restrict__ tr_arow = nullptr, const double* __restrict__ tr_pred = nullptr,
                                                      const double* __restrict__ tr_d = nullptr, const double* __restrict__ tr_b = nullptr) {
    scalars_body<false>(rowpart, m, n, mode, sc, normpart, respart, rank_p, bt, tr_tgt, tr_arow, tr_pred, tr_d, tr_b);
}

// ---- "last workgroup finishes": the single-workgroup / ten-workgroup kernels that sat between the wide kernels of an EM iteration run as
// the TAIL of the kernel in front of them, executed by whichever workgroup of that kernel retires last (a counter in the problem's workspace;
// the last one resets it).  Seven dependent launches per iteration become three -- each launch of the frame loop's match stream waits
// 10-25 us for a workgroup slot beside the U-Net -- with the arithmetic of every piece unchanged (same device functions, same order):
// results are bit-identical to the seven-launch form (tests/test_gpu_match.py compares the two).  Opt-in (CT_EM_FUSE=1): it is SLOWER, see em_fuse().
// Visibility: a workgroup publishes its part with __threadfence() before it takes a ticket; the last one fences again before it reads the
// others' parts (none of which it has touched earlier in the kernel: nothing stale in its L1), and values the tail itself produces are handed
// on in registers / LDS, never re-read from global memory through a possibly cached line.
__device__ __forceinline__ bool em_last_block(int* ticket) {
    __shared__ int em_is_last;
    __threadfence();                                             // every wave: its own stores are complete and written back before the workgroup takes its
    __syncthreads();                                             // ticket (s_barrier does not imply s_waitcnt vmcnt(0) on this target)
    if (threadIdx.x == 0) {
        __threadfence();                                         // ONE release per workgroup: the L2 write-back it performs is cache-wide, not per thread
        const int nblk = (int)gridDim.x;
        const int t = atomicAdd(ticket, 1);
        em_is_last = (t == nblk - 1) ? 1 : 0;
        if (em_is_last) atomicExch(ticket, 0);
    }
    __syncthreads();
    if (em_is_last) __threadfence();
    return em_is_last != 0;
}

// E-step: one wave per target row (trackerlite.py:375-382 / track.py:81-88)
// Is the prior "one value per row, except at most one column"?  (simple_match: 0.1/(n-1) everywhere, 0.9 at the matched column,
// trackerlite.py:242-259.)  One wave per row: lo = the value at least two of the first three columns share; the row is structured
// if at most one entry differs from lo BITWISE.  Table (doubles, sp_m = the batch's largest m): lo[sp_m], hi[sp_m], then idx as
// int32.  A problem with any other row sets dense[problem] and keeps streaming its matrix: the values posterior_kernel uses are
// the matrix's own either way (bit-identical; CT_PRIOR_SCAN=0 always streams).
__global__ __launch_bounds__(256) void prior_scan_kernel(const double* __restrict__ prior, int m, int n, double* __restrict__ sp, int sp_m,
                                                         int* __restrict__ dense, Bt bt) {
    BT_SHIFT(const double*, prior); BT_SHIFT(double*, sp);
    if (bt.dims) { m = bt.dims[4 * blockIdx.z]; n = bt.dims[4 * blockIdx.z + 1]; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= m) return;
    const double* row = prior + (size_t)t * n;
    const long long b0 = __double_as_longlong(row[0]), b1 = __double_as_longlong(row[min(1, n - 1)]), b2 = __double_as_longlong(row[min(2, n - 1)]);
    const long long lo = (b0 == b1 || b0 == b2) ? b0 : b1;
    int cnt = 0, last = -1;
    for (int r = lane; r < n; r += 64)
        if (__double_as_longlong(row[r]) != lo) { ++cnt; last = r; }
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) { cnt += __shfl_xor(cnt, mk); last = max(last, __shfl_xor(last, mk)); }
    if (lane != 0) return;
    if (cnt > 1) { atomicExch(&dense[blockIdx.z], 1); return; }
    sp[t] = __longlong_as_double(lo);
    sp[sp_m + t] = cnt == 1 ? row[last] : __longlong_as_double(lo);
    ((int*)(sp + 2 * (size_t)sp_m))[t] = cnt == 1 ? last : -1;
}
// CT_SIGMA_TRACE=0: sigma2 from one more pass over the posterior (dist2_rowsum_kernel) instead of the trace identity
static bool sigma_trace() { static const bool v = !(getenv("CT_SIGMA_TRACE") && getenv("CT_SIGMA_TRACE")[0] == '0'); return v; }
static bool prior_scan() { static const bool v = !(getenv("CT_PRIOR_SCAN") && getenv("CT_PRIOR_SCAN")[0] == '0'); return v; }

constexpr int PO_REG = 16;                      // posterior_kernel: rows of up to 64 * PO_REG columns are held in registers
__global__ __launch_bounds__(256) void posterior_kernel(const double* __restrict__ prior, const double* __restrict__ pred,
                                                        int n, const double* __restrict__ tgt, int m,
                                                        const double* __restrict__ sc, int legacy, double vol,
                                                        double* __restrict__ P, double s2v = 0.0, double gammav = 0.0,
                                                        Bt bt = Bt{0, nullptr}, const double* __restrict__ sp = nullptr,
                                                        const int* __restrict__ sp_dense = nullptr, int sp_m = 0,
                                                        double* __restrict__ arow = nullptr) {
    BT_SHIFT(const double*, prior); BT_SHIFT(const double*, pred); BT_SHIFT(const double*, tgt); BT_SHIFT(double*, P);
    if (sc) BT_SHIFT(const double*, sc);
    if (sp) BT_SHIFT(const double*, sp);
    if (arow) BT_SHIFT(double*, arow);
    if (bt.dims) { m = bt.dims[4 * blockIdx.z]; n = bt.dims[4 * blockIdx.z + 1]; }
    if (sc && sc[S_DONE] != 0.0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= m) return;
    const double s2 = sc ? sc[S_SIGMA2] : s2v, gamma = sc ? sc[S_GAMMA] : gammav;
    const double two_s2 = 2.0 * s2;
    const double norm = pow(2.0 * M_PI * s2, 1.5);
    // the three per-element divisions of the formula (d2 / 2 sigma2, .. / norm, num / den) have row-invariant divisors; an fp64
    // divide is ~15 instructions (two of them quarter-rate), a third of this loop, so they are multiplications by reciprocals
    // computed once per row (<= 1 ulp per factor, far inside the accumulation-order differences to the numpy formulation):
    // 134 -> 127 us per batched EM iteration
    const double inv_two_s2 = 1.0 / two_s2, coef = (1.0 - gamma) / norm;
    const double yx = tgt[3 * t], yy = tgt[3 * t + 1], yz = tgt[3 * t + 2];
    const double* pr = prior + (size_t)t * n;
    double* po = P + (size_t)t * n;
    double acc = 0.0;
    // a prior built by simple_match is one value per row plus at most one matched column (prior_scan_kernel): its m x n matrix
    // need not be streamed from HBM in every EM iteration
    const bool structured = sp && sp_dense[blockIdx.z] == 0;
    double sp_lo = 0.0, sp_hi = 0.0; int sp_idx = -1;
    if (structured) { sp_lo = sp[t]; sp_hi = sp[sp_m + t]; sp_idx = ((const int*)(sp + 2 * (size_t)sp_m))[t]; }
    if (n <= 64 * PO_REG) {
        // the row's numerators stay in registers until the row sum is known: P is written once (the two-pass form below writes
        // every row twice and reads it back, which made this kernel the largest HBM consumer of an EM iteration).  Same values,
        // same order of the sum.
        double v[PO_REG];
#pragma unroll
        for (int q = 0; q < PO_REG; ++q) {
            if (64 * q >= n) break;
            const int r = lane + 64 * q;
            v[q] = 0.0;
            if (r < n) {
                const double dx = pred[3 * r] - yx, dy = pred[3 * r + 1] - yy, dz = pred[3 * r + 2] - yz;
                const double k = exp(-(dx * dx + dy * dy + dz * dz) * inv_two_s2);
                double prv;
                if (structured) prv = (r == sp_idx) ? sp_hi : sp_lo; else prv = pr[r];
                const double num = legacy ? prv * k : coef * prv * k;
                v[q] = num;
                acc += num;
            }
        }
        acc = wave_sum_d(acc);
        const double den = legacy ? acc + gamma * norm / ((1.0 - gamma) * vol) : acc + gamma / vol;
        const double inv_den = 1.0 / den;
        double asum = 0.0;
#pragma unroll
        for (int q = 0; q < PO_REG; ++q) {
            if (64 * q >= n) break;
            const int r = lane + 64 * q;
            if (r < n) { const double p = v[q] * inv_den; po[r] = p; asum += p; }
        }
        if (arow) { asum = wave_sum_d(asum); if (lane == 0) arow[t] = asum; }
        return;
    }
    for (int r = lane; r < n; r += 64) {
        const double dx = pred[3 * r] - yx, dy = pred[3 * r + 1] - yy, dz = pred[3 * r + 2] - yz;
        const double k = exp(-(dx * dx + dy * dy + dz * dz) * inv_two_s2);
        double prv;
        if (structured) prv = (r == sp_idx) ? sp_hi : sp_lo; else prv = pr[r];
        const double num = legacy ? prv * k : coef * prv * k;
        po[r] = num;
        acc += num;
    }
    acc = wave_sum_d(acc);
    const double den = legacy ? acc + gamma * norm / ((1.0 - gamma) * vol) : acc + gamma / vol;
    const double inv_den = 1.0 / den;
    double asum = 0.0;
    for (int r = lane; r < n; r += 64) { const double p = po[r] * inv_den; po[r] = p; asum += p; }
    if (arow) { asum = wave_sum_d(asum); if (lane == 0) arow[t] = asum; }
}

constexpr int CS_SEG = 32;              // row segments of colstats_kernel (dense / legacy paths and the unfused E-step)
#ifndef CT_ES_SEG
#define CT_ES_SEG 64
#endif
constexpr int ES_SEG = CT_ES_SEG;       // row segments of the fused E-step (32: +0.7 % in the pipelined benchmark, but a single match's EM iteration
                                        // 93 instead of 80 us: a segment's rows are a dependent chain; 128: 77 us, -1.4 %)
constexpr int PART_SEG = ES_SEG > CS_SEG ? ES_SEG : CS_SEG;      // slices the partial-sum buffer is sized for
// column statistics, stage 1: block (x: 64 columns, y: row segment) -> part[seg][4][n] = colsum, Y^T P
__global__ __launch_bounds__(256) void colstats_kernel(const double* __restrict__ P, const double* __restrict__ tgt, int m, int n,
                                                       double* __restrict__ part, const double* __restrict__ sc = nullptr,
                                                       Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, P); BT_SHIFT(const double*, tgt); BT_SHIFT(double*, part);
    if (sc) BT_SHIFT(const double*, sc);
    if (bt.dims) { m = bt.dims[4 * blockIdx.z]; n = bt.dims[4 * blockIdx.z + 1]; }
    if (sc && sc[S_DONE] != 0.0) return;
    __shared__ double red[4][64][4];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + cl, seg = blockIdx.y;
    const int per = (m + CS_SEG - 1) / CS_SEG;
    const int t0 = seg * per, t1 = min(m, t0 + per);
    double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    if (r < n)
        for (int t = t0 + rl; t < t1; t += 4) {
            const double p = P[(size_t)t * n + r];
            s += p; sx = fma(tgt[3 * t], p, sx); sy = fma(tgt[3 * t + 1], p, sy); sz = fma(tgt[3 * t + 2], p, sz);
        }
    red[rl][cl][0] = s; red[rl][cl][1] = sx; red[rl][cl][2] = sy; red[rl][cl][3] = sz;
    __syncthreads();
    if (rl == 0 && r < n) {
        double* o = part + (size_t)seg * 4 * n;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q * n + r] = (red[0][cl][q] + red[1][cl][q]) + (red[2][cl][q] + red[3][cl][q]);
    }
}

// Fused E-step (low-rank EM loop, TrackerLite dialect, n <= 64 * NQ <= 1024): posterior_kernel + colstats_kernel in ONE pass that never
// writes the m x n posterior (unless the caller asked for it).  A wave takes whole target rows t = g, g + 4 CS_SEG, ... (g = its
// global wave index): numerators in registers -> row sum (the same butterfly as posterior_kernel, the same values bit for bit) ->
// normalise -> add p and p * y_t to the lane's per-column accumulators.  The four waves of a block then add their accumulators through
// LDS in a fixed order and the block writes part[seg][colsum | Y^T P][n] -- the layout colstats_finish_par_kernel already reduces.
// Against posterior + colstats this removes one write and one read of the posterior per iteration (2 x 2.9 MB per 600-cell problem)
// and one launch; the column sums are added in another order than colstats_kernel's (rows strided over waves instead of contiguous
// segments), in the single and the batched path alike.  CT_ESTEP_FUSED=0 restores the two-kernel form.
template <int NQ>
__global__ __launch_bounds__(256) void estep_cols_kernel(const double* __restrict__ prior, const double* __restrict__ pred_in, int n,
                                                         const double* __restrict__ tgt, int m, const double* __restrict__ sc, double vol,
                                                         double* __restrict__ P /* or null */, double* __restrict__ part, Bt bt,
                                                         const double* __restrict__ sp, const int* __restrict__ sp_dense, int sp_m,
                                                         double* __restrict__ arow) {
    const double* pred = pred_in;
    BT_SHIFT(const double*, prior); BT_SHIFT(const double*, pred); BT_SHIFT(const double*, tgt); BT_SHIFT(const double*, sc);
    BT_SHIFT(double*, part);
    if (P) BT_SHIFT(double*, P);
    if (sp) BT_SHIFT(const double*, sp);
    if (arow) BT_SHIFT(double*, arow);
    if (bt.dims) { m = bt.dims[4 * blockIdx.z]; n = bt.dims[4 * blockIdx.z + 1]; }
    if (sc[S_DONE] != 0.0) return;
    __shared__ __attribute__((aligned(16))) double red[NQ * 64 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double s2 = sc[S_SIGMA2], gamma = sc[S_GAMMA];
    const double two_s2 = 2.0 * s2;
    const double norm = pow(2.0 * M_PI * s2, 1.5);
    const double inv_two_s2 = 1.0 / two_s2, coef = (1.0 - gamma) / norm;      // (as in posterior_kernel)
    const bool structured = sp && sp_dense[blockIdx.z] == 0;
    double cs[NQ], cx[NQ], cy[NQ], cz[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) cs[q] = cx[q] = cy[q] = cz[q] = 0.0;
    for (int t = blockIdx.x * 4 + wave; t < m; t += 4 * (int)gridDim.x) {
        const double yx = tgt[3 * t], yy = tgt[3 * t + 1], yz = tgt[3 * t + 2];
        const double* pr = prior + (size_t)t * n;

        double sp_lo = 0.0, sp_hi = 0.0; int sp_idx = -1;
        if (structured) { sp_lo = sp[t]; sp_hi = sp[sp_m + t]; sp_idx = ((const int*)(sp + 2 * (size_t)sp_m))[t]; }
        // numerators: a ROLLED loop that parks them in the wave's slice of LDS (the reduction buffer, idle until the rows are done).
        // Unrolled with the numerators in registers the compiler interleaves all NQ fp64 exp chains: 246 VGPRs at NQ = 10, two waves per
        // SIMD, and in the pipelined benchmark those waves wait for half a register file to come free beside the conv waves.
        double* const vq = red + wave * (NQ * 64) + lane;
        double acc = 0.0;
#pragma clang loop unroll(disable)
        for (int q = 0; q < NQ; ++q) {
            const int r = lane + 64 * q;
            double num = 0.0;
            if (r < n) {
                const double dx = pred[3 * r] - yx, dy = pred[3 * r + 1] - yy, dz = pred[3 * r + 2] - yz;
                const double k = exp(-(dx * dx + dy * dy + dz * dz) * inv_two_s2);
                double prv;
                if (structured) prv = (r == sp_idx) ? sp_hi : sp_lo; else prv = pr[r];
                num = coef * prv * k;
                acc += num;
            }
            vq[64 * q] = num;
        }
        acc = wave_sum_d(acc);
        const double inv_den = 1.0 / (acc + gamma / vol);
        double asum = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = lane + 64 * q;
            if (r < n) {
                const double p = vq[64 * q] * inv_den;
                if (P) P[(size_t)t * n + r] = p;
                asum += p;
                cs[q] += p; cx[q] = fma(yx, p, cx[q]); cy[q] = fma(yy, p, cy[q]); cz[q] = fma(yz, p, cz[q]);
            }
        }
        if (arow) { asum = wave_sum_d(asum); if (lane == 0) arow[t] = asum; }
    }
    __syncthreads();                                       // every wave is done with its numerator slice: the buffer becomes the reduction's
    // ((wave 3 + wave 2) + wave 1) + wave 0, through one LDS copy of the accumulators
#pragma unroll
    for (int w = 3; w >= 1; --w) {
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double* d = red + (q * 64 + lane) * 4;
                if (w == 3) { d[0] = cs[q]; d[1] = cx[q]; d[2] = cy[q]; d[3] = cz[q]; }
                else { d[0] += cs[q]; d[1] += cx[q]; d[2] += cy[q]; d[3] += cz[q]; }
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        double* o = part + (size_t)blockIdx.x * 4 * n;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = lane + 64 * q;
            if (r < n) {
                const double* d = red + (q * 64 + lane) * 4;
                o[r] = d[0] + cs[q]; o[n + r] = d[1] + cx[q]; o[2 * n + r] = d[2] + cy[q]; o[3 * n + r] = d[3] + cz[q];
            }
        }
    }
}
// Fused E-step, narrow-wave form: a BLOCK owns a contiguous segment of target rows and walks them one at a time; wave w of the block owns
// the columns [64 NQ w, 64 NQ (w + 1)) of every row, so a lane carries NQ column coordinates, NQ numerators and 4 NQ accumulators (139 VGPRs
// at NQ = 5; estep_cols_kernel's waves need 183-246 and wait for half a register file beside the conv waves of the pipelined benchmark).
// Row sum = per-wave butterfly sums added in wave order through a double-buffered LDS slot (one barrier per
// row); every column belongs to exactly one wave, so the block writes its part[seg][4][n] slice without any cross-wave reduction.
// The row loop is straight-line code: NQ independent exponentials (exp_nonpos) interleave, 52 instructions per pair (72 when every
// pair sat in its own `r < n` branch with the library exp: one match's EM iteration 80 -> 71 us; the pipelined benchmark is unchanged,
// there the kernel's time is the wait for slots between conv workgroups).  MAXT = 512 up to 8 waves (n <= 2560), 1024 beyond (128 VGPRs).
// The posterior's row sums for the sigma2 trace identity are total * 1 / den (the exact sum of the normalised row up to rounding).
__device__ __forceinline__ void colstats_finish_par_body(int vb, const double* part, int n, const double* xref, double* dvec,
                                                         double* sqd, double* rhs, double* braw, int nseg, double (*red)[64][4]);
template <int NQ>
__device__ __forceinline__ void estep_rows_body(const double* __restrict__ prior, const double* pred, int n,
                                                const double* __restrict__ tgt, int m, const double* sc, double vol,
                                                double* __restrict__ P /* or null */, double* part, Bt bt,
                                                const double* __restrict__ sp, const int* __restrict__ sp_dense, int sp_m,
                                                double* __restrict__ arow, const EmOv* ov = nullptr) {
    BT_SHIFT(const double*, prior); BT_SHIFT(const double*, pred); BT_SHIFT(const double*, tgt); BT_SHIFT(const double*, sc);
    BT_SHIFT(double*, part);
    if (P) BT_SHIFT(double*, P);
    if (sp) BT_SHIFT(const double*, sp);
    if (arow) BT_SHIFT(double*, arow);
    if (bt.dims) { m = bt.dims[4 * blockIdx.z]; n = bt.dims[4 * blockIdx.z + 1]; }
    if (!ov && sc[S_DONE] != 0.0) return;
    __shared__ double psum[2][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, W = blockDim.x >> 6;
    const int bx = ov ? ov->vb : (int)blockIdx.x, gx = ov ? ov->nvb : (int)gridDim.x;
    const double s2 = ov ? ov->s2 : sc[S_SIGMA2], gamma = ov ? ov->gamma : sc[S_GAMMA];
    const double two_s2 = 2.0 * s2;
    const double norm = pow(2.0 * M_PI * s2, 1.5);
    const double inv_two_s2 = 1.0 / two_s2, coef = (1.0 - gamma) / norm;
    const bool structured = sp && sp_dense[blockIdx.z] == 0;
    double cs[NQ], cx[NQ], cy[NQ], cz[NQ];
    // this lane's columns do not change from row to row: their coordinates are read once.  A column beyond n gets prior 0, so its
    // numerator is +0.0 and every sum below is what skipping it gives, without a branch around each exponential (the NQ chains of
    // dependent fp64 operations interleave, which is what hides their latency when few waves share the SIMD).
    double px[NQ], py[NQ], pz[NQ];
    bool ok[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        cs[q] = cx[q] = cy[q] = cz[q] = 0.0;
        const int r = (wave * NQ + q) * 64 + lane;
        ok[q] = r < n;
        const int rc = ok[q] ? r : 0;
        px[q] = pred[3 * rc]; py[q] = pred[3 * rc + 1]; pz[q] = pred[3 * rc + 2];
    }
    const int per = (m + gx - 1) / gx;
    const int t0 = bx * per, t1 = min(m, t0 + per);
    // (the row loop exists once per prior form, so that no branch sits between the exponentials)
    auto rows = [&](auto structured_c) {
    constexpr bool STRUCT = decltype(structured_c)::value;
    int buf = 0;
    for (int t = t0; t < t1; ++t, buf ^= 1) {
        const double yx = tgt[3 * t], yy = tgt[3 * t + 1], yz = tgt[3 * t + 2];
        const double* pr = prior + (size_t)t * n;
        double sp_lo = 0.0, sp_hi = 0.0; int sp_idx = -1;
        if (STRUCT) { sp_lo = sp[t]; sp_hi = sp[sp_m + t]; sp_idx = ((const int*)(sp + 2 * (size_t)sp_m))[t]; }
        double v[NQ];
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = (wave * NQ + q) * 64 + lane;
            const double dx = px[q] - yx, dy = py[q] - yy, dz = pz[q] - yz;
            const double k = exp_nonpos(-(dx * dx + dy * dy + dz * dz) * inv_two_s2);
            double prv;
            if (STRUCT) prv = (r == sp_idx) ? sp_hi : sp_lo; else prv = pr[ok[q] ? r : 0];
            prv = ok[q] ? prv : 0.0;
            const double num = coef * prv * k;
            v[q] = num;
            acc += num;
        }
        acc = wave_sum_d(acc);
        if (lane == 0) psum[buf][wave] = acc;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < W; ++w) tot += psum[buf][w];
        const double inv_den = 1.0 / (tot + gamma / vol);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = (wave * NQ + q) * 64 + lane;
            const double p = v[q] * inv_den;
            if (P && ok[q]) P[(size_t)t * n + r] = p;
            cs[q] += p; cx[q] = fma(yx, p, cx[q]); cy[q] = fma(yy, p, cy[q]); cz[q] = fma(yz, p, cz[q]);
        }
        if (arow && threadIdx.x == 0) arow[t] = tot * inv_den;
    }
    };
    if (structured) rows(std::true_type{}); else rows(std::false_type{});
    double* o = part + (size_t)bx * 4 * n;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = (wave * NQ + q) * 64 + lane;
        if (r < n) { o[r] = cs[q]; o[n + r] = cx[q]; o[2 * n + r] = cy[q]; o[3 * n + r] = cz[q]; }
    }
}
template <int NQ, int MAXT>
__global__ __launch_bounds__(MAXT) void estep_rows_kernel(const double* __restrict__ prior, const double* __restrict__ pred, int n,
                                                          const double* __restrict__ tgt, int m, const double* __restrict__ sc, double vol,
                                                          double* __restrict__ P /* or null */, double* __restrict__ part, Bt bt,
                                                          const double* __restrict__ sp, const int* __restrict__ sp_dense, int sp_m,
                                                          double* __restrict__ arow) {
    estep_rows_body<NQ>(prior, pred, n, tgt, m, sc, vol, P, part, bt, sp, sp_dense, sp_m, arow);
}
// fused E-step + (last workgroup) the column statistics' finish for all columns (colstats_finish_par_kernel's body, one 64-column block after
// the other): d, sqrt(d), the scaled right-hand sides and P^T Y.  Single problems only.
struct FinishTail { int* ticket; double* dvec; double* sqd; double* rhs; double* braw; };
template <int NQ, int MAXT>
__global__ __launch_bounds__(MAXT) void estep_rows_finish_kernel(const double* __restrict__ prior, const double* pred, int n,
                                                                 const double* __restrict__ tgt, int m, const double* sc, double vol,
                                                                 double* __restrict__ P /* or null */, double* part,
                                                                 double* __restrict__ arow, FinishTail tl) {
    estep_rows_body<NQ>(prior, pred, n, tgt, m, sc, vol, P, part, Bt{0, nullptr}, nullptr, nullptr, 0, arow);
    if (!em_last_block(tl.ticket)) return;
    if (sc[S_DONE] != 0.0) return;
    __shared__ double red[4][64][4];
    for (int vb = 0; vb * 64 < n; ++vb) {
        colstats_finish_par_body(vb, part, n, pred, tl.dvec, tl.sqd, tl.rhs, tl.braw, (int)gridDim.x, red);
        __syncthreads();                                         // (red is reused by the next block of columns)
    }
}
// CT_ESTEP_FUSED: 0 = posterior + colstats kernels, 1 = estep_cols_kernel (a wave owns whole rows), 2 (default) = estep_rows_kernel
static int estep_mode() { static const int v = getenv("CT_ESTEP_FUSED") ? atoi(getenv("CT_ESTEP_FUSED")) : 2; return v; }
// columns per lane: a CONSTANT, so that wave w always owns columns [320 w, 320 (w + 1)) and the row sums of a problem do not depend on the
// widest problem of its batch (waves beyond a problem's n add +0.0): single and batched runs stay bit-identical.  5 = two waves at n = 600.
static int estep_nq() { static const int v = getenv("CT_ESTEP_NQ") ? atoi(getenv("CT_ESTEP_NQ")) : 5; return v; }
// launches the fused E-step; false if it is switched off or n is too wide for it
static bool launch_estep_cols(int n_max, unsigned zB, hipStream_t st, const double* prior, const double* pred, int n, const double* tgt, int m,
                              const double* sc, double* P, double* part, Bt bt, const double* sp, const int* sp_dense, int sp_m, double* arow) {
    const int need = (n_max + 63) / 64;
    const int mode = estep_mode();
    if (mode <= 0) return false;
    if (mode >= 2) {
        int nq = estep_nq(); if (nq < 1 || nq > 6) nq = 5;
        const int W = (need + nq - 1) / nq;
        if (W <= 16) {
#define CT_ESTEPR(NQv) do { if (W <= 8) CT_ESTEPR2(NQv, 512); else CT_ESTEPR2(NQv, 1024); } while (0)
#define CT_ESTEPR2(NQv, MT) hipLaunchKernelGGL((estep_rows_kernel<NQv, MT>), dim3(ES_SEG, 1, zB), dim3(64 * W), 0, st, prior, pred, n, tgt, m, sc, 1.0, P, \
                                               part, bt, sp, sp_dense, sp_m, arow)
            switch (nq) { case 1: CT_ESTEPR(1); break; case 2: CT_ESTEPR(2); break; case 3: CT_ESTEPR(3); break; case 4: CT_ESTEPR(4); break;
                          case 6: CT_ESTEPR(6); break; default: CT_ESTEPR(5); }
#undef CT_ESTEPR2
#undef CT_ESTEPR
            return true;
        }
    }
    if (need > PO_REG) return false;
#define CT_ESTEP(NQv) hipLaunchKernelGGL(estep_cols_kernel<NQv>, dim3(ES_SEG, 1, zB), dim3(256), 0, st, prior, pred, n, tgt, m, sc, 1.0, P, part, bt, \
                                         sp, sp_dense, sp_m, arow)
    if (need <= 1) CT_ESTEP(1); else if (need <= 2) CT_ESTEP(2); else if (need <= 4) CT_ESTEP(4); else if (need <= 6) CT_ESTEP(6);
    else if (need <= 8) CT_ESTEP(8); else if (need <= 10) CT_ESTEP(10); else if (need <= 12) CT_ESTEP(12); else CT_ESTEP(16);
#undef CT_ESTEP
    return true;
}

// CT_EM_FUSE=1: three launches per EM iteration instead of seven (the "last workgroup finishes" form above).  OFF by default -- measured and
// refuted in round 5: every workgroup has to publish its part with an agent-scope release before it takes its ticket, and on this multi-XCD
// part such a fence writes the XCD's L2 back: one 600-point PR-GLS iteration 160 us alone -> 295-320 us (fence by every thread / by one thread
// per workgroup), the frame loop 6.42 -> 7.30-7.88 ms per frame (profiles/r05_conv_experiments.txt).  Round 2 had refuted fence-based fusion
// of these kernels in another form; kernel boundaries remain the cheapest device-wide release there is.  Bit-identical either way (tested).
static bool em_fuse() { static const bool v = getenv("CT_EM_FUSE") && getenv("CT_EM_FUSE")[0] == '1'; return v; }
// the fused E-step with the column statistics' finish as its tail (single problem); false if this shape takes another E-step form
static bool launch_estep_rows_finish(int n, hipStream_t st, const double* prior, const double* pred, const double* tgt, int m, const double* sc, double* P,
                                     double* part, double* arow, FinishTail tl) {
    const int need = (n + 63) / 64;
    if (estep_mode() < 2) return false;
    int nq = estep_nq(); if (nq < 1 || nq > 6) nq = 5;
    const int W = (need + nq - 1) / nq;
    if (W > 16) return false;
#define CT_ESTEPF(NQv) do { if (W <= 8) CT_ESTEPF2(NQv, 512); else CT_ESTEPF2(NQv, 1024); } while (0)
#define CT_ESTEPF2(NQv, MT) hipLaunchKernelGGL((estep_rows_finish_kernel<NQv, MT>), dim3(ES_SEG), dim3(64 * W), 0, st, prior, pred, n, tgt, m, sc, 1.0, P, part, arow, tl)
    switch (nq) { case 1: CT_ESTEPF(1); break; case 2: CT_ESTEPF(2); break; case 3: CT_ESTEPF(3); break; case 4: CT_ESTEPF(4); break;
                  case 6: CT_ESTEPF(6); break; default: CT_ESTEPF(5); }
#undef CT_ESTEPF2
#undef CT_ESTEPF
    return true;
}

// stage 2: d[r] = colsum, rhs (scaled) and sumP.  xref = points whose X^T D term is subtracted
// (lite: current prediction, trackerlite.py:414; legacy: the original X, track.py:93)
__global__ __launch_bounds__(256) void colstats_finish_kernel(const double* __restrict__ part, int n, const double* __restrict__ xref,
                                                              double lambda, double* __restrict__ sc, double* __restrict__ dvec,
                                                              double* __restrict__ sqd, double* __restrict__ rhs /* [n][3] */) {
    __shared__ double red[4];
    double tot = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) {
        double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
        for (int seg = 0; seg < CS_SEG; ++seg) {
            const double* o = part + (size_t)seg * 4 * n;
            s += o[r]; sx += o[n + r]; sy += o[2 * n + r]; sz += o[3 * n + r];
        }
        dvec[r] = s;
        const double q = sqrt(s);
        sqd[r] = q;
        // b_r = Y^T P[:, r] - x_r d_r ;  scaled rhs = b_r / sqrt(d_r)  (0 when d_r == 0: then b_r == 0 too)
        const double bx = sx - xref[3 * r] * s, by = sy - xref[3 * r + 1] * s, bz = sz - xref[3 * r + 2] * s;
        const double iq = q > 0.0 ? 1.0 / q : 0.0;
        rhs[3 * r] = bx * iq; rhs[3 * r + 1] = by * iq; rhs[3 * r + 2] = bz * iq;
        tot += s;
    }
    tot = wave_sum_d(tot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        sc[S_SUMP] = (red[0] + red[1]) + (red[2] + red[3]);
        sc[S_C] = lambda * sc[S_SIGMA2];
    }
}

// M = D^1/2 G D^1/2 + c I   (lower triangle is what the factorisation reads; fill everything)
__global__ __launch_bounds__(256) void assemble_kernel(const double* __restrict__ G, const double* __restrict__ sqd,
                                                       const double* __restrict__ sc, int n, double* __restrict__ M,
                                                       const double* __restrict__ rhs = nullptr, double* __restrict__ W = nullptr) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (W && gid < 3 * (size_t)n) { const int d = (int)(gid / n), i = (int)(gid - (size_t)d * n); W[gid] = rhs[3 * i + d]; }
    if (gid >= (size_t)n * n) return;
    const int i = (int)(gid / n), j = (int)(gid - (size_t)i * n);
    double v = sqd[i] * G[gid] * sqd[j];
    if (i == j) v += sc[S_C];
    M[gid] = v;
}

// movement of a point set through the field: mov[j][d] = sum_i C[d][i] G[i][j]; one wave per j.
// flags: add (pts += mov), replace (pts = base + mov), accumulate |mov|^2 into norm partials
// Gt is stored [cols][n] (row j contiguous over i): the symmetric n x n Gram matrix as is, and the
// tracked-set kernel computed transposed.  mode 3 = "add unless this is EM iteration 1" (device counter).
__global__ __launch_bounds__(256) void apply_field_kernel(const double* __restrict__ C, const double* __restrict__ Gt, int n, int cols,
                                                          double* __restrict__ pts, const double* __restrict__ base, int mode,
                                                          double* __restrict__ norm_part /* [cols] or null */,
                                                          const double* __restrict__ sc = nullptr,
                                                          const double* __restrict__ dvec = nullptr, const double* __restrict__ sqd = nullptr,
                                                          const double* __restrict__ rhs = nullptr, double* __restrict__ res_part = nullptr,
                                                          Bt bt = Bt{0, nullptr}) {
    if (bt.dims) {                                                         // batched legacy chain: square field on the problem's own reference set
        BT_SHIFT(const double*, C); BT_SHIFT(const double*, Gt); BT_SHIFT(double*, pts); BT_SHIFT(const double*, base);
        n = cols = bt.dims[4 * blockIdx.z + 1];
    }
    if (sc && sc[S_DONE] != 0.0) return;
    if (mode == 3) mode = (sc[S_IT] >= 1.0) ? 1 : 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= cols) return;
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double g = Gt[(size_t)j * n + i];
        ax = fma(C[i], g, ax); ay = fma(C[n + i], g, ay); az = fma(C[2 * n + i], g, az);
    }
    ax = wave_sum_d(ax); ay = wave_sum_d(ay); az = wave_sum_d(az);
    if (lane == 0) {
        if (mode == 1) { pts[3 * j] += ax; pts[3 * j + 1] += ay; pts[3 * j + 2] += az; }
        else if (mode == 2) { pts[3 * j] = base[3 * j] + ax; pts[3 * j + 1] = base[3 * j + 1] + ay; pts[3 * j + 2] = base[3 * j + 2] + az; }
        if (norm_part) norm_part[j] = ax * ax + ay * ay + az * az;
        if (res_part) {   // residual of the ORIGINAL system with the exact Gram matrix: d_j (G C^T)_j + c C_j - b_j
            const double c = sc[S_C], d = dvec[j], q = sqd[j];
            const double bx = q * rhs[3 * j], by = q * rhs[3 * j + 1], bz = q * rhs[3 * j + 2];
            const double rx = d * ax + c * C[j] - bx, ry = d * ay + c * C[n + j] - by, rz = d * az + c * C[2 * n + j] - bz;
            const double rr = fabs(rx) + fabs(ry) + fabs(rz);
            res_part[j] = isfinite(rr) ? fmax(fabs(rx), fmax(fabs(ry), fabs(rz))) : INFINITY;
            res_part[cols + j] = fmax(fabs(bx), fmax(fabs(by), fabs(bz)));
        }
    }
}

// TrackerLite-dialect field application for both point sets in one launch: waves [0, n) move the ref set
// (symmetric Gram matrix G, |movement|^2 partials, exact-system residual monitor), waves [n, n+l) the tracked
// set (kernel stored [l][n]).  Movements are added only from EM iteration 2 on (trackerlite.py:339-341).
__device__ __forceinline__ void apply_dual_body(const double* __restrict__ C, const double* __restrict__ G, int n,
                                                double* predn, const double* __restrict__ Gln, int l,
                                                double* __restrict__ predl, double* norm_part,
                                                const double* sc, const double* dvec,
                                                const double* __restrict__ sqd, const double* __restrict__ rhs,
                                                double* res_part, Bt bt, int first_only, const EmOv* ov = nullptr) {
    BT_SHIFT(const double*, C); BT_SHIFT(const double*, G); BT_SHIFT(double*, predn); BT_SHIFT(const double*, Gln);
    BT_SHIFT(double*, predl); BT_SHIFT(double*, norm_part); BT_SHIFT(const double*, sc); BT_SHIFT(const double*, dvec);
    BT_SHIFT(const double*, sqd); BT_SHIFT(const double*, rhs);
    if (res_part) BT_SHIFT(double*, res_part);
    if (bt.dims) { n = bt.dims[4 * blockIdx.z + 1]; l = bt.dims[4 * blockIdx.z + 2]; }
    if (first_only) l = 0;
    if (!ov && sc[S_DONE] != 0.0) return;
    const bool add = ov ? ov->add : sc[S_IT] >= 1.0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int j = (ov ? ov->vb : (int)blockIdx.x) * 4 + wave;
    if (j >= n + l) return;
    const bool second = j >= n;
    if (second) j -= n;
    const double* row = second ? Gln + (size_t)j * n : G + (size_t)j * n;
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double g = row[i];
        ax = fma(C[i], g, ax); ay = fma(C[n + i], g, ay); az = fma(C[2 * n + i], g, az);
    }
    ax = wave_sum_d(ax); ay = wave_sum_d(ay); az = wave_sum_d(az);
    if (lane != 0) return;
    double* pts = second ? predl : predn;
    // (j is wave-uniform: under ov -- inside em_persistent_kernel, where these arrays change between phases of ONE launch -- such reads
    //  must not become scalar loads)
    auto LD = [&](const double* q) { return ov ? em_fresh_f64(q) : *q; };
    if (add) { pts[3 * j] = LD(pts + 3 * j) + ax; pts[3 * j + 1] = LD(pts + 3 * j + 1) + ay; pts[3 * j + 2] = LD(pts + 3 * j + 2) + az; }
    if (second) return;
    norm_part[j] = ax * ax + ay * ay + az * az;
    if (res_part) {
        const double c = ov ? ov->c : sc[S_C], d = LD(dvec + j), q = LD(sqd + j);
        const double bx = q * LD(rhs + 3 * j), by = q * LD(rhs + 3 * j + 1), bz = q * LD(rhs + 3 * j + 2);
        const double rx = d * ax + c * LD(C + j) - bx, ry = d * ay + c * LD(C + n + j) - by, rz = d * az + c * LD(C + 2 * n + j) - bz;
        const double rr = fabs(rx) + fabs(ry) + fabs(rz);                 // NaN/Inf must not be swallowed by fmax
        res_part[j] = isfinite(rr) ? fmax(fabs(rx), fmax(fabs(ry), fabs(rz))) : INFINITY;
        res_part[n + j] = fmax(fabs(bx), fmax(fabs(by), fabs(bz)));
    }
}
__global__ __launch_bounds__(256) void apply_dual_kernel(const double* __restrict__ C, const double* __restrict__ G, int n,
                                                         double* __restrict__ predn, const double* __restrict__ Gln, int l,
                                                         double* __restrict__ predl, double* __restrict__ norm_part,
                                                         const double* __restrict__ sc, const double* __restrict__ dvec,
                                                         const double* __restrict__ sqd, const double* __restrict__ rhs,
                                                         double* __restrict__ res_part, Bt bt = Bt{0, nullptr}, int first_only = 0) {
    apply_dual_body(C, G, n, predn, Gln, l, predl, norm_part, sc, dvec, sqd, rhs, res_part, bt, first_only);
}
// field application + (last workgroup) the iteration's scalars: sigma2 by the trace identity, gamma, the iteration counter, convergence and
// the low-rank residual monitor (scalars_kernel's body).  Single problems only (no batch table).
struct ScalarsTail { int* ticket; double* sc; int m; int mode; int* rank_p; const double* tgt; const double* arow; const double* trb; const double* rowpart; };
__global__ __launch_bounds__(256) void apply_dual_scalars_kernel(const double* __restrict__ C, const double* __restrict__ G, int n,
                                                                 double* predn, const double* __restrict__ Gln, int l,
                                                                 double* __restrict__ predl, double* norm_part,
                                                                 const double* sc, const double* dvec,
                                                                 const double* __restrict__ sqd, const double* __restrict__ rhs,
                                                                 double* res_part, ScalarsTail tl) {
    apply_dual_body(C, G, n, predn, Gln, l, predl, norm_part, sc, dvec, sqd, rhs, res_part, Bt{0, nullptr}, 0);
    if (!em_last_block(tl.ticket)) return;
    scalars_body<false>(tl.rowpart, tl.m, n, tl.mode, tl.sc, norm_part, res_part, tl.rank_p, Bt{0, nullptr}, tl.tgt, tl.arow, predn, dvec, tl.trb);
}



// ------------------------------------------------------------------------------------------------
// Low-rank fast path of the M-step.  A wide Gaussian Gram matrix is numerically low-rank (rank ~60-90
// for beta = 3 in normalised units, independent of n), so G = U^T U (U: r x n) from a pivoted Cholesky
// computed ONCE per match, and every EM iteration solves only an r x r SPD system:
//   (c I + W W^T)^-1 = (1/c) [ I - W (c I_r + W^T W)^-1 W^T ],   W = D^1/2 U^T  (n x r)
// (Woodbury on the symmetrically scaled system; its cancellation error is cond * eps, the same as a
// direct factorisation's).  The truncation |G - U^T U| <= tol perturbs the solution by ~ tol/c.
// ------------------------------------------------------------------------------------------------
// EM iterations enqueued before the host looks at the convergence flags again: a first short chunk (matches with a good prior converge in
// 6-11 iterations; the rest of a 16-iteration chunk would be ~90 launches that return at once): 6, 6, 8, then 16 at a time (8, 8, 16 until round 5:
// a match that converges in 10 iterations left 42 launches that return at once on the frame loop's match stream, now 14)
static inline int prgls_chunk(int enq, int total) { const int c = enq < 12 ? 6 : (enq < 20 ? 8 : 16); return (total - enq) < c ? (total - enq) : c; }

constexpr int LR_RMAX = 128;
constexpr int EM_TICKET = 4;                   // w.rank[4..6]: tickets of the three fused EM kernels (em_last_block); zeroed per call
constexpr int LR_PF = 8;                       // rows of U fetched together in lowrank_factor_kernel's update loop (general path)
constexpr int LR_REG = 24, LR_PF2 = 16;        // n <= 1024: rows of a thread's column of U kept in registers; rows fetched together beyond them

// pivoted Cholesky of the symmetric PSD matrix G (n x n); single workgroup.
// U [LR_RMAX][n] (row k contiguous).  The factorisation is nested (row k does not depend on later rows), so one run to the fine
// tolerance serves both ranks: rank_out[0] = rows needed for tol_coarse (-1 if not reached within LR_RMAX steps),
// rank_out[1] = rows computed (tol_fine reached, or LR_RMAX).
__global__ __launch_bounds__(1024) void lowrank_factor_kernel(const double* __restrict__ G, int n, double tol_coarse, double tol,
                                                              double* __restrict__ U, double* __restrict__ resid,
                                                              int* __restrict__ rank_out, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, G); BT_SHIFT(double*, U); BT_SHIFT(double*, resid); BT_SHIFT(int*, rank_out);
    if (bt.dims) n = bt.dims[4 * blockIdx.z + 1];
    __shared__ double redv[2][16];
    __shared__ int redi[2][16];
    __shared__ double pivv; __shared__ int pivi;
    __shared__ double urow[LR_RMAX + LR_PF2];   // U[0..k-1][p], zero padded
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n <= 1024) {
        // One point per thread (every match of the per-frame path): the thread's residual and the first LR_REG rows of its column of U stay in
        // REGISTERS -- those rows are read at every later step, and fetching them from L2 was the step's cost (k / LR_PF dependent round trips:
        // 0.4 ms per 600-point factorisation, a fifth of a chained frame's match) --, rows beyond come LR_PF2 at a time, the wave partials of the
        // arg-max are combined by every thread (one barrier less), and thread p hands over its registers for urow.  Same operations in the same
        // order per element as the general loop below.
        const int i = tid; const bool act = i < n;
        double res = act ? G[(size_t)i * n + i] : -1.0;
        double ureg[LR_REG];
#pragma unroll
        for (int j = 0; j < LR_REG; ++j) ureg[j] = 0.0;
        int k = 0, k_coarse = -1;
        for (; k < LR_RMAX; ++k) {
            const bool cand = act && res > -1.0;                  // (the general loop's `d > best` from best = -1)
            double best = cand ? res : -1.0; int bi = cand ? i : 0x7fffffff;
#pragma unroll
            for (int mk = 32; mk >= 1; mk >>= 1) {
                const double od = shfl_xor_d(best, mk); const int oi = __shfl_xor(bi, mk);
                if (od > best || (od == best && oi < bi)) { best = od; bi = oi; }
            }
            double (&rv)[16] = redv[k & 1]; int (&ri)[16] = redi[k & 1];      // (two sets: a wave may run ahead into the next step's partials)
            if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
            __syncthreads();
            double pv = rv[0]; int p = ri[0];
#pragma unroll
            for (int w = 1; w < 16; ++w) { const double b = rv[w]; const int ix = ri[w]; if (b > pv || (b == pv && ix < p)) { pv = b; p = ix; } }
            if (k_coarse < 0 && !(pv > tol_coarse)) k_coarse = k;
            if (!(pv > tol)) break;
            const double piv = sqrt(pv);
            if (tid == p) {
#pragma unroll
                for (int j = 0; j < LR_REG; ++j) urow[j] = ureg[j];          // (rows >= k are still zero)
            }
            if (tid >= LR_REG && tid < LR_RMAX + LR_PF2) urow[tid] = tid < k ? U[(size_t)tid * n + p] : 0.0;
            __syncthreads();
            if (act) {
                double u = G[(size_t)p * n + i];
#pragma unroll
                for (int j = 0; j < LR_REG; ++j) u -= ureg[j] * urow[j];
                for (int j0 = LR_REG; j0 < k; j0 += LR_PF2) {
                    double t[LR_PF2];
#pragma unroll
                    for (int q = 0; q < LR_PF2; ++q) t[q] = U[(size_t)min(j0 + q, k - 1) * n + i];
#pragma unroll
                    for (int q = 0; q < LR_PF2; ++q) u -= t[q] * urow[j0 + q];
                }
                u /= piv;
                U[(size_t)k * n + i] = u;
#pragma unroll
                for (int j = 0; j < LR_REG; ++j) ureg[j] = (j == k) ? u : ureg[j];
                res = (i == p) ? 0.0 : res - u * u;
            }
            __syncthreads();                                   // row k is visible to the threads that fetch U[k][p] next step
        }
        if (act) resid[i] = res;
        if (tid == 0) { rank_out[0] = k_coarse; rank_out[1] = k; }
        return;
    }
    double (&redv0)[16] = redv[0]; int (&redi0)[16] = redi[0];
    for (int i = tid; i < n; i += 1024) resid[i] = G[(size_t)i * n + i];
    __syncthreads();
    int k = 0, k_coarse = -1;
    for (; k < LR_RMAX; ++k) {
        double best = -1.0; int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 1024) { const double d = resid[i]; if (d > best) { best = d; bi = i; } }
#pragma unroll
        for (int mk = 32; mk >= 1; mk >>= 1) {
            const double od = shfl_xor_d(best, mk); const int oi = __shfl_xor(bi, mk);
            if (od > best || (od == best && oi < bi)) { best = od; bi = oi; }
        }
        if (lane == 0) { redv0[wave] = best; redi0[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double b = redv0[0]; int ix = redi0[0];
            for (int w = 1; w < 16; ++w) if (redv0[w] > b || (redv0[w] == b && redi0[w] < ix)) { b = redv0[w]; ix = redi0[w]; }
            pivv = b; pivi = ix;
        }
        __syncthreads();
        if (k_coarse < 0 && !(pivv > tol_coarse)) k_coarse = k;
        if (!(pivv > tol)) break;
        const int p = pivi; const double piv = sqrt(pivv);
        if (tid < LR_RMAX + LR_PF) urow[tid] = tid < k ? U[(size_t)tid * n + p] : 0.0;      // zeros behind row k - 1
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            double u = G[(size_t)p * n + i];              // row p == column p bit for bit ((a - b)^2 == (b - a)^2), and coalesced
            // same subtraction order as the plain loop, but LR_PF loads are in flight together: the plain loop waited one L2 round
            // trip per previous row (k x ~400 cycles, the whole cost of a step); rows past k - 1 are re-reads times the zero padding
            for (int j0 = 0; j0 < k; j0 += LR_PF) {
                double t[LR_PF];
#pragma unroll
                for (int q = 0; q < LR_PF; ++q) t[q] = U[(size_t)min(j0 + q, k - 1) * n + i];
#pragma unroll
                for (int q = 0; q < LR_PF; ++q) u -= t[q] * urow[j0 + q];
            }
            u /= piv;
            U[(size_t)k * n + i] = u;
            resid[i] = (i == p) ? 0.0 : resid[i] - u * u;
        }
        __syncthreads();
    }
    if (tid == 0) { rank_out[0] = k_coarse; rank_out[1] = k; }
}

// column statistics, stage 2 (parallel): 64 columns per block, the CS_SEG segment partials split over
// 4 waves; writes d_i, sqrt d_i and the scaled right-hand side b~_i = (Y^T P[:, i] - x_i d_i) / sqrt d_i.
// body of colstats_finish_par_kernel for the 64 columns of virtual block `vb`, by a workgroup of W waves (the kernel itself: 4; as the tail of the
// fused E-step: that kernel's 2-16): wave w plays the roles of the original's waves w, w + W, ..., so every partial sum has the same terms in
// the same order and the four partials are added in the same tree
__device__ __forceinline__ void colstats_finish_par_body(int vb, const double* part, int n, const double* xref, double* dvec,
                                                         double* sqd, double* rhs, double* braw, int nseg, double (*red)[64][4]) {
    const int cl = threadIdx.x & 63, wv0 = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int i = vb * 64 + cl;
    for (int wv = wv0; wv < 4; wv += W) {
        double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
        if (i < n)
            for (int seg = wv; seg < nseg; seg += 4) {
                const double* o = part + (size_t)seg * 4 * n;
                s += o[i]; sx += o[n + i]; sy += o[2 * n + i]; sz += o[3 * n + i];
            }
        red[wv][cl][0] = s; red[wv][cl][1] = sx; red[wv][cl][2] = sy; red[wv][cl][3] = sz;
    }
    __syncthreads();
    if (wv0 == 0 && i < n) {
        const double s = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
        const double sx = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
        const double sy = (red[0][cl][2] + red[1][cl][2]) + (red[2][cl][2] + red[3][cl][2]);
        const double sz = (red[0][cl][3] + red[1][cl][3]) + (red[2][cl][3] + red[3][cl][3]);
        const double q = sqrt(s);
        const double iq = q > 0.0 ? 1.0 / q : 0.0;
        dvec[i] = s; sqd[i] = q;
        rhs[3 * i] = (sx - xref[3 * i] * s) * iq; rhs[3 * i + 1] = (sy - xref[3 * i + 1] * s) * iq;
        rhs[3 * i + 2] = (sz - xref[3 * i + 2] * s) * iq;
        if (braw) { braw[3 * i] = sx; braw[3 * i + 1] = sy; braw[3 * i + 2] = sz; }      // P^T Y for the sigma2 trace identity (scalars_kernel)
    }
}
__global__ __launch_bounds__(256) void colstats_finish_par_kernel(const double* __restrict__ part, int n, const double* __restrict__ xref,
                                                                  const double* __restrict__ sc, double* __restrict__ dvec,
                                                                  double* __restrict__ sqd, double* __restrict__ rhs, Bt bt = Bt{0, nullptr},
                                                                  double* __restrict__ braw = nullptr, int nseg = CS_SEG) {
    BT_SHIFT(const double*, part); BT_SHIFT(const double*, xref); BT_SHIFT(const double*, sc); BT_SHIFT(double*, dvec);
    BT_SHIFT(double*, sqd); BT_SHIFT(double*, rhs);
    if (braw) BT_SHIFT(double*, braw);
    if (bt.dims) n = bt.dims[4 * blockIdx.z + 1];
    if (sc[S_DONE] != 0.0) return;
    __shared__ double red[4][64][4];
    colstats_finish_par_body((int)blockIdx.x, part, n, xref, dvec, sqd, rhs, braw, nseg, red);
}

// S = U D U^T (r x r, lower triangle) and y = U D^1/2 b~ (r x 3): one wave per entry, lanes stride the n rows
// (coalesced reads of the rows of U, which stay L2-resident); deterministic butterfly reduction.
__global__ __launch_bounds__(256) void lr_gram_kernel(int n, const double* __restrict__ U, const int* __restrict__ rank_p,
                                                      const double* __restrict__ sc, const double* __restrict__ dvec,
                                                      const double* __restrict__ sqd, const double* __restrict__ rhs,
                                                      double* __restrict__ Sout /* [LR_RMAX][LR_RMAX] */, double* __restrict__ yout /* [LR_RMAX][3] */,
                                                      Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, U); BT_SHIFT(const int*, rank_p); BT_SHIFT(const double*, sc); BT_SHIFT(const double*, dvec);
    BT_SHIFT(const double*, sqd); BT_SHIFT(const double*, rhs); BT_SHIFT(double*, Sout); BT_SHIFT(double*, yout);
    if (bt.dims) n = bt.dims[4 * blockIdx.z + 1];
    if (sc[S_DONE] != 0.0) return;
    const int r = *rank_p;
    const int ntri = r * (r + 1) / 2;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= ntri + 3 * r) return;
    double acc = 0.0;
    if (e < ntri) {
        int a2 = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((a2 + 1) * (a2 + 2) / 2 <= e) ++a2;
        while (a2 * (a2 + 1) / 2 > e) --a2;
        const int b2 = e - a2 * (a2 + 1) / 2;
        const double* ua = U + (size_t)a2 * n; const double* ub = U + (size_t)b2 * n;
        for (int i = lane; i < n; i += 64) acc = fma(ua[i] * dvec[i], ub[i], acc);
        acc = wave_sum_d(acc);
        if (lane == 0) Sout[a2 * LR_RMAX + b2] = acc;
    } else {
        const int q = e - ntri, a2 = q / 3, d = q - a2 * 3;
        const double* ua = U + (size_t)a2 * n;
        for (int i = lane; i < n; i += 64) acc = fma(ua[i] * sqd[i], rhs[3 * i + d], acc);
        acc = wave_sum_d(acc);
        if (lane == 0) yout[a2 * 3 + d] = acc;
    }
}

// The same sums, 4 x 4 entries per wave (lr_gram_kernel re-reads two rows of U per ENTRY: 31 MB through the caches per 600-point
// problem and the third-largest kernel of a batched chain).  Bit-identical by construction: every lane accumulates the same
// i = lane (mod 64) terms in the same order, and the 16 butterflies run "transposed" (at each step a lane keeps half of its
// accumulators and hands the other half to its partner; fp add is commutative, so the kept sums equal wave_sum_d's bit for bit).
constexpr int LG_T = 4;
constexpr int LG_UN = 1;                       // i-steps whose loads are in flight together (4: -0.5 % in the match-bound pipeline)
__device__ __forceinline__ void wave_sum16_d(double (&acc)[16], int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int mk = 32 >> s;
        const bool hi = (lane & mk) != 0;
#pragma unroll
        for (int k = 0; k < (8 >> s); ++k) {
            const double a = acc[2 * k], b = acc[2 * k + 1];
            const double send = hi ? a : b, keep = hi ? b : a;
            acc[k] = keep + shfl_xor_d(send, mk);
        }
    }
    acc[0] += shfl_xor_d(acc[0], 2);
    acc[0] += shfl_xor_d(acc[0], 1);              // lane holds entry q = 8 bit2 + 4 bit3 + 2 bit4 + bit5 (bits of its lane number)
}
__device__ __forceinline__ void lr_gram_tiled_body(int n, const double* __restrict__ U, const int* __restrict__ rank_p,
                                                   const double* sc, const double* dvec,
                                                   const double* sqd, const double* rhs,
                                                   double* Sout, double* yout, Bt bt, const EmOv* ov = nullptr) {
    BT_SHIFT(const double*, U); BT_SHIFT(const int*, rank_p); BT_SHIFT(const double*, sc); BT_SHIFT(const double*, dvec);
    BT_SHIFT(const double*, sqd); BT_SHIFT(const double*, rhs); BT_SHIFT(double*, Sout); BT_SHIFT(double*, yout);
    if (bt.dims) n = bt.dims[4 * blockIdx.z + 1];
    if (!ov && sc[S_DONE] != 0.0) return;
    const int r = ov ? ov->r : *rank_p;
    const int T = (r + LG_T - 1) / LG_T, ntri = T * (T + 1) / 2;
    const int w = (ov ? ov->vb : (int)blockIdx.x) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= ntri + T) return;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    const int q_mine = ((lane >> 2) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
    if (w < ntri) {
        int ta = (int)((sqrt(8.0 * w + 1.0) - 1.0) * 0.5);
        while ((ta + 1) * (ta + 2) / 2 <= w) ++ta;
        while (ta * (ta + 1) / 2 > w) --ta;
        const int tb = w - ta * (ta + 1) / 2;
        const int a0 = ta * LG_T, b0 = tb * LG_T;
        const double* ua[LG_T]; const double* ub[LG_T];
#pragma unroll
        for (int k = 0; k < LG_T; ++k) { ua[k] = U + (size_t)min(a0 + k, r - 1) * n; ub[k] = U + (size_t)min(b0 + k, r - 1) * n; }
        // LG_UN steps' operands are fetched together (the wave's time is its chain of load round trips, not its 160 fma);
        // a step past the end loads a clamped index and skips its fma, so the terms and their order stay the same
        for (int i0 = lane; i0 < n; i0 += 64 * LG_UN) {
            double dd[LG_UN], va[LG_UN][LG_T], vb[LG_UN][LG_T];
#pragma unroll
            for (int u = 0; u < LG_UN; ++u) {
                const int i = min(i0 + 64 * u, n - 1);
                dd[u] = dvec[i];
#pragma unroll
                for (int k = 0; k < LG_T; ++k) { va[u][k] = ua[k][i]; vb[u][k] = ub[k][i]; }
            }
#pragma unroll
            for (int u = 0; u < LG_UN; ++u) {
                if (i0 + 64 * u < n) {
#pragma unroll
                    for (int k = 0; k < LG_T; ++k) {
                        const double wa = va[u][k] * dd[u];
#pragma unroll
                        for (int j = 0; j < LG_T; ++j) acc[k * LG_T + j] = fma(wa, vb[u][j], acc[k * LG_T + j]);
                    }
                }
            }
        }
        wave_sum16_d(acc, lane);
        const int a2 = a0 + q_mine / LG_T, b2 = b0 + q_mine % LG_T;
        if ((lane & 3) == 0 && a2 < r && b2 <= a2) Sout[a2 * LR_RMAX + b2] = acc[0];
    } else {
        const int a0 = (w - ntri) * LG_T;
        const double* ua[LG_T];
#pragma unroll
        for (int k = 0; k < LG_T; ++k) ua[k] = U + (size_t)min(a0 + k, r - 1) * n;
        for (int i0 = lane; i0 < n; i0 += 64 * LG_UN) {
            double qq[LG_UN], bb[LG_UN][3], va[LG_UN][LG_T];
#pragma unroll
            for (int u = 0; u < LG_UN; ++u) {
                const int i = min(i0 + 64 * u, n - 1);
                qq[u] = sqd[i]; bb[u][0] = rhs[3 * i]; bb[u][1] = rhs[3 * i + 1]; bb[u][2] = rhs[3 * i + 2];
#pragma unroll
                for (int k = 0; k < LG_T; ++k) va[u][k] = ua[k][i];
            }
#pragma unroll
            for (int u = 0; u < LG_UN; ++u) {
                if (i0 + 64 * u < n) {
#pragma unroll
                    for (int k = 0; k < LG_T; ++k) {
                        const double wa = va[u][k] * qq[u];
                        acc[k * LG_T] = fma(wa, bb[u][0], acc[k * LG_T]); acc[k * LG_T + 1] = fma(wa, bb[u][1], acc[k * LG_T + 1]);
                        acc[k * LG_T + 2] = fma(wa, bb[u][2], acc[k * LG_T + 2]);
                    }
                }
            }
        }
        wave_sum16_d(acc, lane);
        const int a2 = a0 + q_mine / LG_T, d = q_mine % LG_T;
        if ((lane & 3) == 0 && a2 < r && d < 3) yout[a2 * 3 + d] = acc[0];
    }
}
__global__ __launch_bounds__(256) void lr_gram_tiled_kernel(int n, const double* __restrict__ U, const int* __restrict__ rank_p,
                                                            const double* __restrict__ sc, const double* __restrict__ dvec,
                                                            const double* __restrict__ sqd, const double* __restrict__ rhs,
                                                            double* __restrict__ Sout, double* __restrict__ yout, Bt bt = Bt{0, nullptr}) {
    lr_gram_tiled_body(n, U, rank_p, sc, dvec, sqd, rhs, Sout, yout, bt);
}
// CT_GRAM_TILED=0 selects the entry-per-wave kernel (A/B and the bit-identity test)
static bool gram_tiled() { static const bool v = !(getenv("CT_GRAM_TILED") && getenv("CT_GRAM_TILED")[0] == '0'); return v; }

// ------------------------------------------------------------------------------------------------
// Row-group variant of apply_dual_kernel for the batched TrackerLite chain (RG rows per wave).
// The wave-per-row kernel re-reads the whole coefficient set (C: 14.4 KB at n = 600) for every 4.8-KB row of G it streams, i.e.
// three quarters of its load instructions and of its L2 traffic are re-reads, and its three butterflies cost as much as its
// loop.  With RG = 4 rows per wave the C loads are shared and the 12 reductions run transposed (wave_sum16_d).  Every lane still
// accumulates the same terms in the same order and the butterfly tree is the same: results are bit-identical to the
// wave-per-row kernel (tests/test_gpu_match.py; CT_ROW_GROUPS=0 selects that one).  B = 16 x 600 points: 29 -> 19 us per launch,
// 150 -> 140 us per batched EM iteration.  (The same treatment of posterior / dist2_rowsum was measured and dropped: their exp /
// distance chains need the occupancy of one row per wave - 36 -> 42 us and 18 -> 21 us.)  Single matches keep one row per wave.
// ------------------------------------------------------------------------------------------------
constexpr int RG = 4;
static bool row_groups() { static const bool v = !(getenv("CT_ROW_GROUPS") && getenv("CT_ROW_GROUPS")[0] == '0'); return v; }
constexpr int RG_MIN_BATCH = 4;                // fewer problems per launch: one row per wave fills the chip better
// lane that holds entry q after wave_sum16_d
__device__ __forceinline__ int sum16_lane(int q) { return (((q >> 3) & 1) << 2) | (((q >> 2) & 1) << 3) | (((q >> 1) & 1) << 4) | ((q & 1) << 5); }

__global__ __launch_bounds__(256) void apply_dual_rg_kernel(const double* __restrict__ C, const double* __restrict__ G, int n,
                                                            double* __restrict__ predn, const double* __restrict__ Gln, int l,
                                                            double* __restrict__ predl, double* __restrict__ norm_part,
                                                            const double* __restrict__ sc, const double* __restrict__ dvec,
                                                            const double* __restrict__ sqd, const double* __restrict__ rhs,
                                                            double* __restrict__ res_part, Bt bt, int first_only = 0) {
    BT_SHIFT(const double*, C); BT_SHIFT(const double*, G); BT_SHIFT(double*, predn); BT_SHIFT(const double*, Gln);
    BT_SHIFT(double*, predl); BT_SHIFT(double*, norm_part); BT_SHIFT(const double*, sc); BT_SHIFT(const double*, dvec);
    BT_SHIFT(const double*, sqd); BT_SHIFT(const double*, rhs);
    if (res_part) BT_SHIFT(double*, res_part);
    if (bt.dims) { n = bt.dims[4 * blockIdx.z + 1]; l = bt.dims[4 * blockIdx.z + 2]; }
    if (first_only) l = 0;
    if (sc[S_DONE] != 0.0) return;
    const bool add = sc[S_IT] >= 1.0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gn = (n + RG - 1) / RG, gl = (l + RG - 1) / RG;
    int g = blockIdx.x * 4 + wave;
    if (g >= gn + gl) return;
    const bool second = g >= gn;
    if (second) g -= gn;
    const int rows = second ? l : n;
    const int j0 = g * RG;
    const double* base = second ? Gln : G;
    const double* row[RG];
#pragma unroll
    for (int k = 0; k < RG; ++k) row[k] = base + (size_t)min(j0 + k, rows - 1) * n;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double cx = C[i], cy = C[n + i], cz = C[2 * n + i];
        double gv[RG];
#pragma unroll
        for (int k = 0; k < RG; ++k) gv[k] = row[k][i];
#pragma unroll
        for (int k = 0; k < RG; ++k) {
            acc[4 * k] = fma(cx, gv[k], acc[4 * k]); acc[4 * k + 1] = fma(cy, gv[k], acc[4 * k + 1]);
            acc[4 * k + 2] = fma(cz, gv[k], acc[4 * k + 2]);
        }
    }
    wave_sum16_d(acc, lane);                       // lane sum16_lane(4 k + d) holds component d of row j0 + k
    const int q_mine = ((lane >> 2) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
    const int k = q_mine >> 2;
    const double ax = acc[0];
    const double ay = shfl_d(acc[0], sum16_lane(4 * k + 1)), az = shfl_d(acc[0], sum16_lane(4 * k + 2));
    const int j = j0 + k;
    if ((lane & 3) != 0 || (q_mine & 3) != 0 || j >= rows) return;
    double* pts = second ? predl : predn;
    if (add) { pts[3 * j] += ax; pts[3 * j + 1] += ay; pts[3 * j + 2] += az; }
    if (second) return;
    norm_part[j] = ax * ax + ay * ay + az * az;
    if (res_part) {
        const double c = sc[S_C], d = dvec[j], q = sqd[j];
        const double bx = q * rhs[3 * j], by = q * rhs[3 * j + 1], bz = q * rhs[3 * j + 2];
        const double rx = d * ax + c * C[j] - bx, ry = d * ay + c * C[n + j] - by, rz = d * az + c * C[2 * n + j] - bz;
        const double rr = fabs(rx) + fabs(ry) + fabs(rz);                 // NaN/Inf must not be swallowed by fmax
        res_part[j] = isfinite(rr) ? fmax(fabs(rx), fmax(fabs(ry), fabs(rz))) : INFINITY;
        res_part[n + j] = fmax(fabs(bx), fmax(fabs(by), fabs(bz)));
    }
}

// The tracked set only rides along (trackerlite.py:335-341: predicted_l += C . G_nl with a FIXED G_nl, from the second iteration on),
// nothing in the EM reads it.  By linearity its final position is tracked + (sum of those C) . G_nl: the batched chain sums the
// coefficients (lr_coeff_kernel) and streams the l x n matrix ONCE here instead of once per iteration (a sixth of an iteration's
// bytes).  Same terms, added in another order: the tracked coordinates differ from the per-iteration form by ~2e-11 in normalised
// units - the cancellation error eps |C| |G| that either form carries (|C| ~ 1e5) -
// (CT_DEFER_TRACKED=0 keeps the per-iteration form; the EM state, iteration counts and posteriors are untouched either way).
__global__ __launch_bounds__(256) void apply_tracked_kernel(const double* __restrict__ Csum, const double* __restrict__ Gln, int n,
                                                            double* __restrict__ predl, int l, const unsigned char* __restrict__ skip, Bt bt) {
    if (skip && skip[blockIdx.z]) return;
    BT_SHIFT(const double*, Csum); BT_SHIFT(const double*, Gln); BT_SHIFT(double*, predl);
    if (bt.dims) { n = bt.dims[4 * blockIdx.z + 1]; l = bt.dims[4 * blockIdx.z + 2]; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= l) return;
    const double* row = Gln + (size_t)j * n;
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double g = row[i];
        ax = fma(Csum[i], g, ax); ay = fma(Csum[n + i], g, ay); az = fma(Csum[2 * n + i], g, az);
    }
    ax = wave_sum_d(ax); ay = wave_sum_d(ay); az = wave_sum_d(az);
    if (lane == 0) { predl[3 * j] += ax; predl[3 * j + 1] += ay; predl[3 * j + 2] += az; }
}
static bool defer_tracked() { static const bool v = !(getenv("CT_DEFER_TRACKED") && getenv("CT_DEFER_TRACKED")[0] == '0'); return v; }

constexpr int LS_B = 8;
// In-LDS solve of the SPD system S q = y for 3 right-hand sides stored as rows r..r+2 of S (augmented Cholesky,
// 8-column blocks, 2 barriers per block; then a blocked back-substitution).  256 threads.  On return rows r..r+2 hold q.
__device__ void chol_factor_aug_lds(double* S, int ld, int r, int naug, int tid) {
    // LDS reads are issued unconditionally (indices clamped into the matrix, padding selected afterwards): predicated
    // reads compile to branches and serialise at ~120 cycles apiece.  Arithmetic and its order are unchanged.
    const int ra = r + naug;
    const int ty = tid >> 4, tx = tid & 15;
    for (int j0 = 0; j0 < r; j0 += LS_B) {
        const int nb = min(LS_B, r - j0);
        // ---- (1) diagonal block factor in registers (every thread), then this thread's panel rows
        double Ld[LS_B][LS_B];
#pragma unroll
        for (int i = 0; i < LS_B; ++i)
#pragma unroll
            for (int k = 0; k < LS_B; ++k) {
                if (k <= i) {
                    const double v = S[min(j0 + i, r - 1) * ld + j0 + min(k, nb - 1)];
                    Ld[i][k] = (i < nb) ? v : (i == k ? 1.0 : 0.0);
                } else Ld[i][k] = 0.0;
            }
#pragma unroll
        for (int k = 0; k < LS_B; ++k) {
            const double dk = sqrt(Ld[k][k]);
            Ld[k][k] = dk;
            const double ik = 1.0 / dk;
#pragma unroll
            for (int i = k + 1; i < LS_B; ++i) Ld[i][k] *= ik;
#pragma unroll
            for (int i = k + 1; i < LS_B; ++i)
#pragma unroll
                for (int q = k + 1; q <= i; ++q) Ld[i][q] -= Ld[i][k] * Ld[q][k];
        }
        for (int a2 = j0 + nb + tid; a2 < ra; a2 += 256) {
            double x[LS_B];
#pragma unroll
            for (int k = 0; k < LS_B; ++k) { const double v = S[a2 * ld + j0 + min(k, nb - 1)]; x[k] = (k < nb) ? v : 0.0; }
#pragma unroll
            for (int k = 0; k < LS_B; ++k) {
                double t = x[k];
#pragma unroll
                for (int q = 0; q < k; ++q) t -= x[q] * Ld[k][q];
                x[k] = t / Ld[k][k];
            }
#pragma unroll
            for (int k = 0; k < LS_B; ++k) if (k < nb) S[a2 * ld + j0 + k] = x[k];
        }
        __syncthreads();
        // ---- (2) trailing rank-nb update; thread 0 also stores the factored diagonal block
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < LS_B; ++i)
#pragma unroll
                for (int k = 0; k < LS_B; ++k) if (i < nb && k <= i) S[(j0 + i) * ld + j0 + k] = Ld[i][k];
        }
        const int base = j0 + nb;
        for (int a2 = base + ty; a2 < ra; a2 += 16) {
            double La[LS_B];
#pragma unroll
            for (int k = 0; k < LS_B; ++k) { const double v = S[a2 * ld + j0 + min(k, nb - 1)]; La[k] = (k < nb) ? v : 0.0; }
            const int bmax = a2 < r ? a2 : r - 1;
            for (int b2 = base + tx; b2 <= bmax; b2 += 16) {
                double Lb[LS_B];
#pragma unroll
                for (int k = 0; k < LS_B; ++k) { const double v = S[b2 * ld + j0 + min(k, nb - 1)]; Lb[k] = (k < nb) ? v : 0.0; }
                const double cur = S[a2 * ld + b2];
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < LS_B; ++k) t = fma(La[k], Lb[k], t);
                S[a2 * ld + b2] = cur - t;
            }
        }
        __syncthreads();
    }
}
__device__ void chol_backsub3_lds(double* S, int ld, int r, int tid) {
    for (int j0 = ((r - 1) / LS_B) * LS_B; j0 >= 0; j0 -= LS_B) {
        const int nb = min(LS_B, r - j0);
        // this step's inputs, all reads in flight together (see chol_factor_aug_lds)
        double Lb[LS_B][LS_B], rh[LS_B][3], dg[LS_B];
#pragma unroll
        for (int k = 0; k < LS_B; ++k) {
            const int ck = j0 + min(k, nb - 1);
            rh[k][0] = S[(r + 0) * ld + ck]; rh[k][1] = S[(r + 1) * ld + ck]; rh[k][2] = S[(r + 2) * ld + ck];
            dg[k] = S[ck * ld + ck];
#pragma unroll
            for (int p2 = k + 1; p2 < LS_B; ++p2) Lb[p2][k] = S[(j0 + min(p2, nb - 1)) * ld + ck];
        }
        double qv[LS_B][3];
#pragma unroll
        for (int k = LS_B - 1; k >= 0; --k) {
            double t0 = rh[k][0], t1 = rh[k][1], t2 = rh[k][2];
#pragma unroll
            for (int p2 = k + 1; p2 < LS_B; ++p2) {
                const double lpk = (p2 < nb) ? Lb[p2][k] : 0.0;        // (x - 0 * q == x exactly: same value as skipping)
                t0 -= lpk * qv[p2][0]; t1 -= lpk * qv[p2][1]; t2 -= lpk * qv[p2][2];
            }
            const double idk = 1.0 / dg[k];
            const bool live = k < nb;
            qv[k][0] = live ? t0 * idk : 0.0; qv[k][1] = live ? t1 * idk : 0.0; qv[k][2] = live ? t2 * idk : 0.0;
        }
        __syncthreads();                                   // every thread has read this step's inputs
        if (tid < j0) {
            double l[LS_B];
#pragma unroll
            for (int k = 0; k < LS_B; ++k) { const double v = S[(j0 + min(k, nb - 1)) * ld + tid]; l[k] = (k < nb) ? v : 0.0; }
            double u0 = S[(r + 0) * ld + tid], u1 = S[(r + 1) * ld + tid], u2 = S[(r + 2) * ld + tid];
#pragma unroll
            for (int k = 0; k < LS_B; ++k) { u0 -= l[k] * qv[k][0]; u1 -= l[k] * qv[k][1]; u2 -= l[k] * qv[k][2]; }
            S[(r + 0) * ld + tid] = u0; S[(r + 1) * ld + tid] = u1; S[(r + 2) * ld + tid] = u2;
        } else if (tid == j0) {
#pragma unroll
            for (int k = 0; k < LS_B; ++k)
                if (k < nb) { S[(r + 0) * ld + j0 + k] = qv[k][0]; S[(r + 1) * ld + j0 + k] = qv[k][1]; S[(r + 2) * ld + j0 + k] = qv[k][2]; }
        }
        __syncthreads();
    }
}

// Low-latency variants for the r x r low-rank system (lr_solve_kernel only; the dense paths keep the division-based
// helpers above).  The critical path of the blocked factorisation is a chain of dependent fp64 sqrt and divide
// sequences (~32 per 8-column block step); here each pivot costs ONE rsqrt (d = x * rsqrt(x)), the reciprocal pivots
// are kept (idiag, LDS) and every later division by a pivot becomes a multiplication.  Differences to the division
// form are a few ulp per entry; the low-rank M-step is verified against the exact Gram matrix every iteration anyway.
__device__ __forceinline__ double rsqrt_nr(double x) {      // v_rsq_f64 (~2^-26) + one Newton step: a 5-op dependent chain
    const double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * y;
    const double e = fma(-x * y, h, 0.5);                    // 0.5 - x y^2 / 2
    return fma(y, e, y);
}
// S is addressed through SIX(row, col): PACK = false -> row * ld + col (odd ld spreads rows over the banks); PACK = true -> the rows of the lower
// triangle back to back (row a holds a + 1 entries; the naug right-hand-side rows, r entries each, follow): half the LDS -- 35 instead of 68 KB
// at rank 90 -- which is what lets the single workgroup of lr_solve_kernel start beside four resident conv workgroups of a CU as soon as ONE of
// them retires (with 68 KB it waited for two to retire together: 137 us per launch in the pipelined benchmark against 22 alone).  Every access
// of the two routines has col <= row (rows < r) or col < r (right-hand-side rows); the arithmetic is untouched, so results keep their bits.
#define SIX(row, col) (PACK ? (((row) < r ? (row) * ((row) + 1) / 2 : r * (r + 1) / 2 + ((row) - r) * r) + (col)) : ((row) * ld + (col)))
template <bool PACK>
__device__ void chol_factor_aug_lds_fast(double* S, double* idiag, int ld, int r, int naug, int tid) {
    // All LDS reads of a phase are issued unconditionally (indices clamped into the matrix, padding selected afterwards)
    // so that they pipeline: predicated reads compile to branches and serialise at ~120 cycles apiece.  Dependent fp64
    // chains are the critical path (~25 cycles per op): explicit fma, split accumulators, one trailing entry per thread.
    const int ra = r + naug;
    for (int j0 = 0; j0 < r; j0 += LS_B) {
        const int nb = min(LS_B, r - j0);
        double Ld[LS_B][LS_B], inv[LS_B];
#pragma unroll
        for (int i = 0; i < LS_B; ++i)
#pragma unroll
            for (int k = 0; k < LS_B; ++k) {
                if (k <= i) {
                    const double v = S[SIX(min(j0 + i, r - 1), j0 + min(k, nb - 1))];
                    Ld[i][k] = (i < nb) ? v : (i == k ? 1.0 : 0.0);
                } else Ld[i][k] = 0.0;
            }
#pragma unroll
        for (int k = 0; k < LS_B; ++k) {
            const double ik = rsqrt_nr(Ld[k][k]);
            inv[k] = ik;
            Ld[k][k] = Ld[k][k] * ik;
#pragma unroll
            for (int i = k + 1; i < LS_B; ++i) Ld[i][k] *= ik;
#pragma unroll
            for (int i = k + 1; i < LS_B; ++i)
#pragma unroll
                for (int q = k + 1; q <= i; ++q) Ld[i][q] = fma(-Ld[i][k], Ld[q][k], Ld[i][q]);
        }
        for (int a2 = j0 + nb + tid; a2 < ra; a2 += 256) {
            double x[LS_B];
#pragma unroll
            for (int k = 0; k < LS_B; ++k) { const double v = S[SIX(a2, j0 + min(k, nb - 1))]; x[k] = (k < nb) ? v : 0.0; }
#pragma unroll
            for (int k = 0; k < LS_B; ++k) {
                double t = x[k];
#pragma unroll
                for (int q = 0; q < k; ++q) t = fma(-x[q], Ld[k][q], t);
                x[k] = t * inv[k];
            }
#pragma unroll
            for (int k = 0; k < LS_B; ++k) if (k < nb) S[SIX(a2, j0 + k)] = x[k];
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < LS_B; ++i) {
                if (i < nb) idiag[j0 + i] = inv[i];
#pragma unroll
                for (int k = 0; k < LS_B; ++k) if (i < nb && k <= i) S[SIX((j0 + i), j0 + k)] = Ld[i][k];
            }
        }
        // trailing update S[a][b] -= L[a][:] . L[b][:] over the lower triangle of the remaining rows (b <= a, b < r) plus
        // the full augmented rows: entries are enumerated linearly so every thread gets at most ceil(count / 256) of them
        const int base = j0 + nb;
        const int nt = r - base;                               // remaining matrix rows
        const int ntri = nt * (nt + 1) / 2;
        const int nent = ntri + naug * nt;
        for (int e = tid; e < nent; e += 256) {
            int ia, ib;
            if (e < ntri) {
                ia = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                while (ia * (ia + 1) / 2 > e) --ia;
                while ((ia + 1) * (ia + 2) / 2 <= e) ++ia;
                ib = e - ia * (ia + 1) / 2;
            } else { const int q = e - ntri; ia = nt + q / nt; ib = q - (q / nt) * nt; }
            const int a2 = base + ia, b2 = base + ib;
            double La[LS_B], Lb[LS_B];
#pragma unroll
            for (int k = 0; k < LS_B; ++k) {
                const int ck = j0 + min(k, nb - 1);
                const double va = S[SIX(a2, ck)];
                La[k] = (k < nb) ? va : 0.0;
                Lb[k] = S[SIX(b2, ck)];
            }
            const double cur = S[SIX(a2, b2)];
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int k = 0; k < LS_B; k += 2) { t0 = fma(La[k], Lb[k], t0); t1 = fma(La[k + 1], Lb[k + 1], t1); }
            S[SIX(a2, b2)] = cur - (t0 + t1);
        }
        __syncthreads();
    }
}
template <bool PACK>
__device__ void chol_backsub3_lds_fast(double* S, const double* idiag, int ld, int r, int tid) {
    for (int j0 = ((r - 1) / LS_B) * LS_B; j0 >= 0; j0 -= LS_B) {
        const int nb = min(LS_B, r - j0);
        // this step's inputs, all reads in flight together: diagonal block (lower triangle), 3 right-hand sides, pivots
        double Lb[LS_B][LS_B], rh[LS_B][3], idg[LS_B];
#pragma unroll
        for (int k = 0; k < LS_B; ++k) {
            const int ck = j0 + min(k, nb - 1);
            rh[k][0] = S[SIX((r + 0), ck)]; rh[k][1] = S[SIX((r + 1), ck)]; rh[k][2] = S[SIX((r + 2), ck)];
            idg[k] = idiag[ck];
#pragma unroll
            for (int p2 = k + 1; p2 < LS_B; ++p2) Lb[p2][k] = S[SIX((j0 + min(p2, nb - 1)), ck)];
        }
        double qv[LS_B][3];
#pragma unroll
        for (int k = LS_B - 1; k >= 0; --k) {
            double t0 = rh[k][0], t1 = rh[k][1], t2 = rh[k][2];
#pragma unroll
            for (int p2 = k + 1; p2 < LS_B; ++p2) {                    // qv[p2] = 0 for p2 >= nb
                t0 = fma(-Lb[p2][k], qv[p2][0], t0); t1 = fma(-Lb[p2][k], qv[p2][1], t1); t2 = fma(-Lb[p2][k], qv[p2][2], t2);
            }
            const bool live = k < nb;
            qv[k][0] = live ? t0 * idg[k] : 0.0; qv[k][1] = live ? t1 * idg[k] : 0.0; qv[k][2] = live ? t2 * idg[k] : 0.0;
        }
        __syncthreads();                                   // every thread has read this step's inputs
        if (tid < j0) {
            double l[LS_B];
#pragma unroll
            for (int k = 0; k < LS_B; ++k) l[k] = S[SIX((j0 + min(k, nb - 1)), tid)];
            double u0 = S[SIX((r + 0), tid)], u1 = S[SIX((r + 1), tid)], u2 = S[SIX((r + 2), tid)];
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
#pragma unroll
            for (int k = 0; k < LS_B; k += 2) {                        // qv = 0 beyond nb
                u0 = fma(-l[k], qv[k][0], u0); u1 = fma(-l[k], qv[k][1], u1); u2 = fma(-l[k], qv[k][2], u2);
                v0 = fma(-l[k + 1], qv[k + 1][0], v0); v1 = fma(-l[k + 1], qv[k + 1][1], v1); v2 = fma(-l[k + 1], qv[k + 1][2], v2);
            }
            S[SIX((r + 0), tid)] = u0 + v0; S[SIX((r + 1), tid)] = u1 + v1; S[SIX((r + 2), tid)] = u2 + v2;
        } else if (tid == j0) {
#pragma unroll
            for (int k = 0; k < LS_B; ++k)
                if (k < nb) { S[SIX((r + 0), j0 + k)] = qv[k][0]; S[SIX((r + 1), j0 + k)] = qv[k][1]; S[SIX((r + 2), j0 + k)] = qv[k][2]; }
        }
        __syncthreads();
    }
}
#undef SIX

// ---- dense path for n > DS_MAXN: blocked right-looking Cholesky of M = D^1/2 G D^1/2 + c I (NB = 32) --------------
// The 3 right-hand sides W [3][n] ride through the factorisation as extra rows (forward substitution for free);
// per panel: (1) every block factors the 32 x 32 diagonal block in LDS with the register-blocked routine above and
// solves its 32 panel rows (the last block: the rhs rows), (2) trailing rank-32 update incl. the rhs rows; then one
// blocked backward substitution kernel.
constexpr int NB = 32;

__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ M, double* __restrict__ W, int n, int k0) {
    __shared__ double Lkk[NB * (NB + 1)];
    __shared__ double Arow[NB][NB + 1];
    const int tid = threadIdx.x;
    const int nb = min(NB, n - k0);
    constexpr int ld = NB + 1;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        Lkk[i * ld + j] = (i < nb && j < nb) ? M[(size_t)(k0 + i) * n + k0 + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    chol_factor_aug_lds(Lkk, ld, nb, 0, tid);
    __syncthreads();
    if (blockIdx.x == 0)
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e % NB;
            if (i < nb && j < nb) M[(size_t)(k0 + i) * n + k0 + j] = (j <= i) ? Lkk[i * ld + j] : 0.0;
        }
    const bool rhs_block = blockIdx.x == gridDim.x - 1;
    const int row0 = k0 + nb + blockIdx.x * NB;
    const int nr = rhs_block ? 3 : min(NB, n - row0);
    if (!rhs_block && row0 >= n) return;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        double v = 0.0;
        if (i < nr && j < nb) v = rhs_block ? W[(size_t)i * n + k0 + j] : M[(size_t)(row0 + i) * n + k0 + j];
        Arow[i][j] = v;
    }
    __syncthreads();
    if (tid < nr) {       // x L_kk^T = a : forward substitution with the row in registers, L broadcast from LDS
        double x[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) x[j] = Arow[tid][j];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double t = x[j];
#pragma unroll
            for (int p2 = 0; p2 < j; ++p2) t = fma(-x[p2], Lkk[j * ld + p2], t);
            x[j] = t / Lkk[j * ld + j];                 // rows/cols >= nb are identity padding
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) Arow[tid][j] = x[j];
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        if (i < nr && j < nb) {
            if (rhs_block) W[(size_t)i * n + k0 + j] = Arow[i][j];
            else M[(size_t)(row0 + i) * n + k0 + j] = Arow[i][j];
        }
    }
}

// trailing update: A[i][j] -= sum_p L[i][k0+p] L[j][k0+p] (lower tiles), rhs rows: W[d][j] -= sum_p W[d][k0+p] L[j][k0+p]
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ M, double* __restrict__ W, int n, int k0) {
    const int nb = min(NB, n - k0);
    const int base = k0 + nb;
    const int ntile = (n - base + NB - 1) / NB;
    const int bi = blockIdx.y, bj = blockIdx.x;
    const bool rhs_tile = bi == ntile;
    if (!rhs_tile && bj > bi) return;
    const int i0 = base + bi * NB, j0 = base + bj * NB;
    if (j0 >= n || (!rhs_tile && i0 >= n)) return;
    __shared__ double Li[NB][NB + 1], Lj[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, p2 = e % NB;
        double vi = 0.0;
        if (p2 < nb) {
            if (rhs_tile) { if (r < 3) vi = W[(size_t)r * n + k0 + p2]; }
            else if (i0 + r < n) vi = M[(size_t)(i0 + r) * n + k0 + p2];
        }
        Li[r][p2] = vi;
        Lj[r][p2] = (j0 + r < n && p2 < nb) ? M[(size_t)(j0 + r) * n + k0 + p2] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = ty + 8 * q, c = tx;
        const int gj = j0 + c;
        if (gj >= n) continue;
        if (rhs_tile) {
            if (r >= 3) continue;
        } else if (i0 + r >= n || gj > i0 + r) continue;
        double acc = 0.0;
#pragma unroll 8
        for (int p2 = 0; p2 < NB; ++p2) acc = fma(Li[r][p2], Lj[c][p2], acc);
        if (rhs_tile) W[(size_t)r * n + gj] -= acc;
        else M[(size_t)(i0 + r) * n + gj] -= acc;
    }
}

// backward substitution L^T z = w for the 3 right-hand sides in W [3][n] (in place), one workgroup, right-looking:
// solve the 32 unknowns of block kb (in-LDS blocked triangle), then every earlier unknown p < k0 subtracts the 32
// terms L[k0+i][p] z_i (rows of L: coalesced over p, no reduction).  Finally C[d][i] = sqrt(d_i) z[d][i].
__global__ __launch_bounds__(256) void chol_backward_kernel(const double* __restrict__ L, int n, double* __restrict__ W,
                                                             const double* __restrict__ sqd, double* __restrict__ C) {
    __shared__ double T[(NB + 3) * (NB + 1)];
    constexpr int ld = NB + 1;
    const int tid = threadIdx.x;
    const int nblk = (n + NB - 1) / NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * NB, nb = min(NB, n - k0);
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e % NB;
            T[i * ld + j] = (i < nb && j < nb) ? L[(size_t)(k0 + i) * n + k0 + j] : (i == j ? 1.0 : 0.0);
        }
        if (tid < 96) { const int c = tid & 31, d = tid >> 5; T[(nb + d) * ld + c] = (c < nb) ? W[(size_t)d * n + k0 + c] : 0.0; }
        __syncthreads();
        chol_backsub3_lds(T, ld, nb, tid);
        if (tid < 96) { const int c = tid & 31, d = tid >> 5; if (c < nb) W[(size_t)d * n + k0 + c] = T[(nb + d) * ld + c]; }
        for (int p2 = tid; p2 < k0; p2 += 256) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll 8
            for (int i = 0; i < NB; ++i) {
                if (i < nb) {
                    const double l = L[(size_t)(k0 + i) * n + p2];
                    a0 = fma(l, T[(nb + 0) * ld + i], a0); a1 = fma(l, T[(nb + 1) * ld + i], a1); a2 = fma(l, T[(nb + 2) * ld + i], a2);
                }
            }
            W[p2] -= a0; W[(size_t)n + p2] -= a1; W[2 * (size_t)n + p2] -= a2;
        }
        __syncthreads();
    }
    for (int e = tid; e < 3 * n; e += 256) { const int i = e % n; C[e] = sqd[i] * W[e]; }
}

// single workgroup: q = (c I + S)^-1 y for 3 right-hand sides, S given by lr_gram_kernel.
// The right-hand sides ride along as 3 extra rows of the matrix being factorised
// ([[S, y], [y^T, .]] = L_aug L_aug^T puts w = L^-1 y into those rows), so the forward substitution is
// free.  Blocked right-looking factorisation, 8 columns per step, 2 barriers per step: (1) every thread
// factors the 8 x 8 diagonal block redundantly in registers and solves its own rows of the panel,
// (2) rank-8 update of the trailing block.  Then a back-substitution, one unknown per thread.
// body of lr_solve_kernel (256 threads, dynamic LDS = packed triangle + 3 right-hand sides).  qs_out (LDS, [3 r]) and c_out: the solution and
// c = lambda sigma2 handed to a caller that goes on in the same workgroup (the fused kernel's coefficient pass) without re-reading global memory
__device__ __forceinline__ void lr_solve_body(const double* Sin, const double* yin, int n, const int* rank_p, double lambda,
                                              const double* dvec, double* sc, double* qout, double* qs_out, double* c_out, const EmOv* ov = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int r = ov ? ov->r : *rank_p;
    const int ld = r | 1;                  // (unpacked form: odd leading dimension)
    double* S = sm;                        // packed: rows of the lower triangle back to back, then the 3 right-hand-side rows (PSIX)
#define PSIX(row, col) (((row) < r ? (row) * ((row) + 1) / 2 : r * (r + 1) / 2 + ((row) - r) * r) + (col))
    __shared__ double red[4];
    __shared__ double idiag[LR_RMAX];
    const int tid = threadIdx.x;
    const double c = lambda * (ov ? ov->s2 : sc[S_SIGMA2]);
    // loads in batches of 8 per thread, all in flight together (a runtime-bound loop would serialise the L2 round trips)
    const float rinv = 1.0f / (float)r;
    for (int e0 = 0; e0 < r * r; e0 += 256 * 8) {
        double v[8]; int ai[8], bi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + tid + 256 * u;
            int a = (int)(((float)e + 0.5f) * rinv);
            a = (a * r > e) ? a - 1 : ((a + 1) * r <= e ? a + 1 : a);
            const int b = e - a * r;
            ai[u] = a; bi[u] = b;
            v[u] = (e < r * r && b <= a) ? Sin[a * LR_RMAX + b] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + tid + 256 * u < r * r && bi[u] <= ai[u]) S[PSIX(ai[u], bi[u])] = v[u] + (ai[u] == bi[u] ? c : 0.0);
    }
    for (int e = tid; e < r * 3; e += 256) { const int a = e / 3, d = e - a * 3; S[PSIX(r + d, a)] = yin[e]; }
    {   // sumP = sum_i d_i
        double acc = 0.0;
        for (int i0 = 0; i0 < n; i0 += 256 * 4) {
            double dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + tid + 256 * u; dv[u] = i < n ? dvec[i] : 0.0; }
            acc += (dv[0] + dv[1]) + (dv[2] + dv[3]);
        }
        acc = wave_sum_d(acc);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
    }
    __syncthreads();
    if (tid == 0) { sc[S_SUMP] = (red[0] + red[1]) + (red[2] + red[3]); sc[S_C] = c; }
    chol_factor_aug_lds_fast<true>(S, idiag, ld, r, 3, tid);
    chol_backsub3_lds_fast<true>(S, idiag, ld, r, tid);
    for (int e = tid; e < r * 3; e += 256) {
        const int a2 = e / 3, d = e - a2 * 3;
        const double qv = S[PSIX(r + d, a2)];
        qout[e] = qv;
        if (qs_out) qs_out[e] = qv;
    }
    if (c_out) *c_out = c;
#undef PSIX
}
__global__ __launch_bounds__(256) void lr_solve_kernel(const double* __restrict__ Sin, const double* __restrict__ yin,
                                                       int n, const int* __restrict__ rank_p, double lambda,
                                                       const double* __restrict__ dvec, double* __restrict__ sc,
                                                       double* __restrict__ qout, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, Sin); BT_SHIFT(const double*, yin); BT_SHIFT(const int*, rank_p); BT_SHIFT(const double*, dvec);
    BT_SHIFT(double*, sc); BT_SHIFT(double*, qout);
    if (bt.dims) n = bt.dims[4 * blockIdx.z + 1];
    if (sc[S_DONE] != 0.0) return;
    lr_solve_body(Sin, yin, n, rank_p, lambda, dvec, sc, qout, nullptr, nullptr);
}

// Dense M-step for small n (n + 3 rows of n|1 doubles fit the LDS: n <= DS_MAXN): one workgroup finishes the column
// statistics, assembles M = D^1/2 G D^1/2 + c I in LDS, solves it with the augmented Cholesky and writes
// C[d][i] = sqrt(d_i) z_i[d].  Replaces finish + assemble + 2 n/32 panel/update launches + the triangular solves.
constexpr int DS_MAXN = 132;
__global__ __launch_bounds__(256) void dense_small_solve_kernel(const double* __restrict__ part, int n, const double* __restrict__ xref,
                                                                const double* __restrict__ G, double lambda, double* __restrict__ sc,
                                                                double* __restrict__ dvec, double* __restrict__ sqd_g,
                                                                double* __restrict__ rhs_g, double* __restrict__ C, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(const double*, part); BT_SHIFT(const double*, xref); BT_SHIFT(const double*, G); BT_SHIFT(double*, sc); BT_SHIFT(double*, dvec);
    BT_SHIFT(double*, sqd_g); BT_SHIFT(double*, rhs_g); BT_SHIFT(double*, C); BT_DIM_N(n);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double sq[DS_MAXN];
    __shared__ double idiag[DS_MAXN];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int ld = n | 1;
    double* S = sm;                         // [n + 3][ld]
    const double c = lambda * sc[S_SIGMA2];
    double tot = 0.0;
    for (int i = tid; i < n; i += 256) {
        double s0 = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
        for (int seg = 0; seg < CS_SEG; ++seg) {
            const double* o = part + (size_t)seg * 4 * n;
            s0 += o[i]; sx += o[n + i]; sy += o[2 * n + i]; sz += o[3 * n + i];
        }
        const double q = sqrt(s0), iq = q > 0.0 ? 1.0 / q : 0.0;
        const double bx = (sx - xref[3 * i] * s0) * iq, by = (sy - xref[3 * i + 1] * s0) * iq, bz = (sz - xref[3 * i + 2] * s0) * iq;
        sq[i] = q; dvec[i] = s0; sqd_g[i] = q;
        rhs_g[3 * i] = bx; rhs_g[3 * i + 1] = by; rhs_g[3 * i + 2] = bz;
        S[(n + 0) * ld + i] = bx; S[(n + 1) * ld + i] = by; S[(n + 2) * ld + i] = bz;
        tot += s0;
    }
    tot = wave_sum_d(tot);
    if ((tid & 63) == 0) red[tid >> 6] = tot;
    __syncthreads();
    if (tid == 0) { sc[S_SUMP] = (red[0] + red[1]) + (red[2] + red[3]); sc[S_C] = c; }
    // assembly of the scaled lower triangle: batches of 8 Gram entries per thread in flight together
    const float ninv = 1.0f / (float)n;
    for (int e0 = 0; e0 < n * n; e0 += 256 * 8) {
        double gv[8]; int ii[8], jj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + tid + 256 * u;
            int i = (int)(((float)e + 0.5f) * ninv);
            i = (i * n > e) ? i - 1 : ((i + 1) * n <= e ? i + 1 : i);
            ii[u] = i; jj[u] = e - i * n;
            gv[u] = (e < n * n && jj[u] <= i) ? G[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + tid + 256 * u < n * n && jj[u] <= ii[u]) S[ii[u] * ld + jj[u]] = sq[ii[u]] * gv[u] * sq[jj[u]] + (ii[u] == jj[u] ? c : 0.0);
    }
    __syncthreads();
    chol_factor_aug_lds_fast<false>(S, idiag, ld, n, 3, tid);
    chol_backsub3_lds_fast<false>(S, idiag, ld, n, tid);
    for (int e = tid; e < 3 * n; e += 256) { const int d = e / n, i = e - d * n; C[e] = sq[i] * S[(n + d) * ld + i]; }
}

// C[d][i] = sqrt(d_i) (b~_i - sqrt(d_i) sum_a U[a][i] q[a]) / c ; 64 rows i per block, the rank split over 4 waves
// body of lr_coeff_kernel for the 64 columns of virtual block `vb`: qs = the solution [3 r] in LDS, c = lambda sigma2
__device__ __forceinline__ void lr_coeff_body(int vb, const double* U, int n, int r, const double* qs, const double* sqd, const double* rhs, double c,
                                              bool count_it, double* C, double* Csum, double (*ps)[64][3]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = vb * 64 + lane;
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (i < n)
        for (int a = wv; a < r; a += 4) {
            const double u = U[(size_t)a * n + i];
            ax = fma(u, qs[3 * a], ax); ay = fma(u, qs[3 * a + 1], ay); az = fma(u, qs[3 * a + 2], az);
        }
    ps[wv][lane][0] = ax; ps[wv][lane][1] = ay; ps[wv][lane][2] = az;
    __syncthreads();
    if (wv == 0 && i < n) {
        const double s = sqd[i];
        ax = (ps[0][lane][0] + ps[1][lane][0]) + (ps[2][lane][0] + ps[3][lane][0]);
        ay = (ps[0][lane][1] + ps[1][lane][1]) + (ps[2][lane][1] + ps[3][lane][1]);
        az = (ps[0][lane][2] + ps[1][lane][2]) + (ps[2][lane][2] + ps[3][lane][2]);
        const double c0 = s * (rhs[3 * i] - s * ax) / c, c1 = s * (rhs[3 * i + 1] - s * ay) / c, c2 = s * (rhs[3 * i + 2] - s * az) / c;
        C[i] = c0; C[n + i] = c1; C[2 * n + i] = c2;
        // deferred field application for the tracked set (apply_tracked_kernel): the coefficients of every iteration whose movement
        // counts (from the second on, trackerlite.py:339-341) are summed; this kernel returns early once the problem has converged
        if (Csum && count_it) { Csum[i] += c0; Csum[n + i] += c1; Csum[2 * n + i] += c2; }
    }
}
__global__ __launch_bounds__(256) void lr_coeff_kernel(const double* __restrict__ U, int n, const int* __restrict__ rank_p,
                                                       const double* __restrict__ q, const double* __restrict__ sqd,
                                                       const double* __restrict__ rhs, const double* __restrict__ sc,
                                                       double* __restrict__ C, Bt bt = Bt{0, nullptr}, double* __restrict__ Csum = nullptr) {
    BT_SHIFT(const double*, U); BT_SHIFT(const int*, rank_p); BT_SHIFT(const double*, q); BT_SHIFT(const double*, sqd);
    BT_SHIFT(const double*, rhs); BT_SHIFT(const double*, sc); BT_SHIFT(double*, C);
    if (Csum) BT_SHIFT(double*, Csum);
    if (bt.dims) n = bt.dims[4 * blockIdx.z + 1];
    if (sc[S_DONE] != 0.0) return;
    __shared__ double ps[4][64][3];
    __shared__ double qs[LR_RMAX * 3];
    const int r = *rank_p;
    for (int e = threadIdx.x; e < 3 * r; e += 256) qs[e] = q[e];
    __syncthreads();
    lr_coeff_body((int)blockIdx.x, U, n, r, qs, sqd, rhs, sc[S_C], sc[S_IT] >= 1.0, C, Csum, ps);
}
// S = U D U^T and y (lr_gram_tiled_kernel's body) + (last workgroup) the r x r solve and the coefficients C of all columns: lr_solve_kernel's and
// lr_coeff_kernel's bodies.  Launched with lr_solve_kernel's dynamic LDS.  Single problems only.
struct SolveTail { int* ticket; double lambda; double* sc; double* q; double* C; };
__global__ __launch_bounds__(256) void lr_gram_solve_coeff_kernel(int n, const double* __restrict__ U, const int* __restrict__ rank_p,
                                                                  const double* sc, const double* dvec, const double* sqd, const double* rhs,
                                                                  double* Sout, double* yout, SolveTail tl) {
    lr_gram_tiled_body(n, U, rank_p, sc, dvec, sqd, rhs, Sout, yout, Bt{0, nullptr});
    if (!em_last_block(tl.ticket)) return;
    if (sc[S_DONE] != 0.0) return;
    __shared__ double ps[4][64][3];
    __shared__ double qs[LR_RMAX * 3];
    __shared__ double c_sh;
    const bool count_it = sc[S_IT] >= 1.0;
    lr_solve_body(Sout, yout, n, rank_p, tl.lambda, dvec, tl.sc, tl.q, qs, &c_sh);
    __syncthreads();
    const int r = *rank_p;
    const double c = c_sh;
    for (int vb = 0; vb * 64 < n; ++vb) {
        lr_coeff_body(vb, U, n, r, qs, sqd, rhs, c, count_it, tl.C, nullptr, ps);
        __syncthreads();                                         // (ps is reused by the next block of columns)
    }
}

// ------------------------------------------------------------------------------------------------
// A chunk of EM iterations of ONE match in ONE launch (the frame loop's match: beside the U-Net each of an iteration's seven launches
// waits 10-25 us for a workgroup slot, 175 us per iteration against 72 alone; a prior that needs 48 iterations made the match stream the
// frame loop's critical path at 11 ms per frame).  A small persistent grid walks the phases -- the SAME device functions the seven kernels
// are made of, called for virtual blocks -- with gd_grid_barrier between them: E-step | finish | Gram | solve (replicated in every
// workgroup: the r x r system is tiny, and its solution stays in LDS for the coefficients) + coefficients | field application | scalars
// (workgroup 0).  Bit-identical to the seven-launch form (tests/test_gpu_match.py).  Between phases of one launch everything travels
// through L2: the barrier is release + acquire, and what a wave reads at a uniform address (the scalar block, a row's own statistics)
// goes through agent-scope loads (EmOv / em_fresh_*): plain ones may be served stale by the scalar cache.
// ------------------------------------------------------------------------------------------------
struct EmP {
    const double* prior; const double* tgt; int m, n, l; double lambda;
    double* predn; double* predl; const double* G; const double* Gln; double* P; double* part; double* dvec; double* sqd; double* rhs;
    double* trb; double* tra; const double* U; int* rank; double* Spart; double* ypart; double* q; double* C;
    double* normpart; double* respart; double* rowpart; double* sc; int* bar; int iters;
};
constexpr int EMP_NQ = 5, EMP_W = 4;            // E-step columns per lane / waves per workgroup: 1280 columns (launch_estep_cols' own NQ)
__global__ __launch_bounds__(256) void em_persistent_kernel(const EmP a) {
    __shared__ double red_ps[4 * 64 * 4];                        // the finish's partials and the coefficients' partials are never live together
    double (*red)[64][4] = reinterpret_cast<double (*)[64][4]>(red_ps);
    double (*ps)[64][3] = reinterpret_cast<double (*)[64][3]>(red_ps);
    __shared__ double qs[LR_RMAX * 3];
    __shared__ double c_sh;
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int n = a.n, m = a.m, l = a.l;
    int passed = 0;
    for (int it = 0; it < a.iters; ++it) {
        if (em_fresh_f64(a.sc + S_DONE) != 0.0) break;           // (every workgroup reads the word after the same barrier: the same value)
        EmOv ov{};
        ov.s2 = em_fresh_f64(a.sc + S_SIGMA2); ov.gamma = em_fresh_f64(a.sc + S_GAMMA); ov.add = em_fresh_f64(a.sc + S_IT) >= 1.0;
        ov.r = em_fresh_i32(a.rank); ov.c = a.lambda * ov.s2;
        // ---- E-step (estep_rows_kernel<5, .>: ES_SEG row segments)
        ov.nvb = ES_SEG;
        for (int vb = bid; vb < ES_SEG; vb += G) {
            ov.vb = vb;
            estep_rows_body<EMP_NQ>(a.prior, a.predn, n, a.tgt, m, a.sc, 1.0, a.P, a.part, Bt{0, nullptr}, nullptr, nullptr, 0, a.tra, &ov);
            __syncthreads();                                     // (the row sums' LDS slots are reused by the next segment)
        }
        gd_grid_barrier(a.bar, G, passed);
        // ---- column statistics' finish (colstats_finish_par_kernel)
        for (int vb = bid; vb * 64 < n; vb += G) {
            colstats_finish_par_body(vb, a.part, n, a.predn, a.dvec, a.sqd, a.rhs, a.trb, ES_SEG, red);
            __syncthreads();
        }
        gd_grid_barrier(a.bar, G, passed);
        // ---- S = U D U^T, y (lr_gram_tiled_kernel)
        {
            const int T = (ov.r + LG_T - 1) / LG_T, nwave = T * (T + 1) / 2 + T;
            for (int vb = bid; vb * 4 < nwave; vb += G) {
                ov.vb = vb;
                lr_gram_tiled_body(n, a.U, a.rank, a.sc, a.dvec, a.sqd, a.rhs, a.Spart, a.ypart, Bt{0, nullptr}, &ov);
            }
        }
        gd_grid_barrier(a.bar, G, passed);
        // ---- the r x r solve in every workgroup (lr_solve_kernel's body; all write the same sumP, c and q), then this workgroup's coefficients
        lr_solve_body(a.Spart, a.ypart, n, a.rank, a.lambda, a.dvec, a.sc, a.q, qs, &c_sh, &ov);
        __syncthreads();
        for (int vb = bid; vb * 64 < n; vb += G) {
            lr_coeff_body(vb, a.U, n, ov.r, qs, a.sqd, a.rhs, c_sh, ov.add, a.C, nullptr, ps);
            __syncthreads();
        }
        gd_grid_barrier(a.bar, G, passed);
        // ---- field application (apply_dual_kernel)
        for (int vb = bid; vb * 4 < n + l; vb += G) {
            ov.vb = vb;
            apply_dual_body(a.C, a.G, n, a.predn, a.Gln, l, a.predl, a.normpart, a.sc, a.dvec, a.sqd, a.rhs, a.respart, Bt{0, nullptr}, 0, &ov);
        }
        gd_grid_barrier(a.bar, G, passed);
        // ---- sigma2 (trace identity), gamma, iteration counter, convergence, residual monitor (scalars_kernel), by workgroup 0
        if (bid == 0)
            scalars_body<true>(a.rowpart, m, n, 1, a.sc, a.normpart, a.respart, a.rank, Bt{0, nullptr}, a.tgt, a.tra, a.predn, a.dvec, a.trb);
        gd_grid_barrier(a.bar, G, passed);
    }
}

// tracker.py:1269-1289: pred[j] += sum_i C[:, i] exp(-|pred_j - inter_i|^2 / 2 beta^2); one wave per j
__global__ __launch_bounds__(256) void gram_apply_kernel(double* __restrict__ pred, int l, const double* __restrict__ inter, int n,
                                                         const double* __restrict__ C, double two_b2, Bt bt = Bt{0, nullptr}) {
    BT_SHIFT(double*, pred); BT_SHIFT(const double*, inter); BT_SHIFT(const double*, C); BT_DIM_N(n);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= l) return;
    const double px = pred[3 * j], py = pred[3 * j + 1], pz = pred[3 * j + 2];
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double dx = px - inter[3 * i], dy = py - inter[3 * i + 1], dz = pz - inter[3 * i + 2];
        const double g = exp(-(dx * dx + dy * dy + dz * dz) / two_b2);
        ax = fma(C[i], g, ax); ay = fma(C[n + i], g, ay); az = fma(C[2 * n + i], g, az);
    }
    ax = wave_sum_d(ax); ay = wave_sum_d(ay); az = wave_sum_d(az);
    if (lane == 0) { pred[3 * j] = px + ax; pred[3 * j + 1] = py + ay; pred[3 * j + 2] = pz + az; }
}

// scipy.stats.trim_mean(stack, cut, axis=0): sort the k values, drop int(cut*k) at both ends
constexpr int TM_MAXK = 64;
__global__ __launch_bounds__(256) void trim_mean_kernel(const double* __restrict__ stack, int k, int n3, int lo,
                                                        double* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n3) return;
    double v[TM_MAXK];
    for (int q = 0; q < k; ++q) v[q] = stack[(size_t)q * n3 + e];
    for (int a = 1; a < k; ++a) {
        const double x = v[a]; int b = a - 1;
        while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
        v[b + 1] = x;
    }
    double s = 0.0;
    for (int q = lo; q < k - lo; ++q) s += v[q];
    out[e] = s / (double)(k - 2 * lo);
}

}  // namespace

// ---- helpers of the batched legacy chain (problem = blockIdx.z, everything at a fixed slab stride)
__global__ __launch_bounds__(256) void bt_copy3_kernel(double* __restrict__ dst, const double* __restrict__ src, int which /* 1: n, 2: l */, Bt bt) {
    BT_SHIFT(double*, dst); BT_SHIFT(const double*, src);
    const int cnt = 3 * bt.dims[4 * blockIdx.z + which];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void bt_zero_kernel(unsigned char* __restrict__ p, int nwords, Bt bt) {
    BT_SHIFT(unsigned char*, p);
    uint32_t* w = reinterpret_cast<uint32_t*>(p);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nwords; i += gridDim.x * 256) w[i] = 0u;
}
__global__ void bt_legacy_init_kernel(double* __restrict__ sc, double* __restrict__ C, Bt bt) {       // gamma0 = 0.1 (track.py:41), C = 0
    BT_SHIFT(double*, sc); BT_SHIFT(double*, C);
    const int n = bt.dims[4 * blockIdx.z + 1];
    if (threadIdx.x < S_NUM) sc[threadIdx.x] = threadIdx.x == S_GAMMA ? 0.1 : 0.0;
    for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) C[i] = 0.0;
}
__global__ void bt_gather_ctr_kernel(const int* __restrict__ ctr, size_t stride, int B, int* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { const int* c = (const int*)((const double*)ctr + (size_t)b * stride); for (int k = 0; k < 4; ++k) out[4 * b + k] = c[k]; }
}
__global__ __launch_bounds__(256) void bt_pack_kernel(const double* __restrict__ src, size_t stride, int cnt, double* __restrict__ dst) {
    const double* sp = src + (size_t)blockIdx.z * stride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) dst[(size_t)blockIdx.z * cnt + i] = sp[i];
}

struct ct_ffn {
    int device;
    float* d_w;          // arena: w1 | bn1[4][512] | w2 | bn2[4][512] | w3
    float b3;
    size_t o_w1, o_bn1, o_w2, o_bn2, o_w3;
};

extern "C" {

// ------------------------------------------------------------------------------------------------ kNN
int ct_normalize_points(const double* points, int n, const double* apply_para, double* out_points, double* para, ct_stream_t stream) {
    if (!points || n <= 0 || (!apply_para && !para)) return CT_EINVAL;
    hipLaunchKernelGGL(normalize_points_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, points, n, apply_para, out_points, para);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_denormalize_points(const double* points, int n, const double* para, double* out_points, ct_stream_t stream) {
    if (!points || !para || !out_points || n <= 0) return CT_EINVAL;
    hipLaunchKernelGGL(denormalize_points_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, points, n, para, out_points);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_knn_features(const double* points, int n, int k, float* feat, ct_stream_t stream) {
    if (!points || !feat || n <= 0 || k <= 0) return CT_EINVAL;
    if (n < k + 1 || n > KNN_MAXN || k + 1 > 32) return CT_ESHAPE;     // sklearn raises when n < k+1
    hipLaunchKernelGGL(knn_features_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, points, n, k, feat);
    LAUNCH_CHECK();
    return CT_OK;
}

// ------------------------------------------------------------------------------------------------ FFN
size_t ct_ffn_num_weights(void) { return (size_t)FEAT * HID + 4 * HID + (size_t)2 * HID * HID + 4 * HID + HID + 1; }

int ct_ffn_create(const float* w, size_t n_floats, int device, ct_ffn_t** out) {
    if (!w || !out) return CT_EINVAL;
    if (n_floats != ct_ffn_num_weights()) return CT_ESHAPE;
    HIPCHK(hipSetDevice(device));
    ct_ffn* h = new (std::nothrow) ct_ffn();
    if (!h) return CT_EINVAL;
    h->device = device;
    h->o_w1 = 0; h->o_bn1 = h->o_w1 + (size_t)FEAT * HID; h->o_w2 = h->o_bn1 + 4 * HID;
    h->o_bn2 = h->o_w2 + (size_t)2 * HID * HID; h->o_w3 = h->o_bn2 + 4 * HID;
    h->b3 = w[n_floats - 1];
    hipError_t e = hipMalloc((void**)&h->d_w, (n_floats - 1) * sizeof(float));
    if (e != hipSuccess) { delete h; return (int)e; }
    e = hipMemcpy(h->d_w, w, (n_floats - 1) * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->d_w); delete h; return (int)e; }
    *out = h;
    return CT_OK;
}

void ct_ffn_destroy(ct_ffn_t* h) { if (h) { (void)hipFree(h->d_w); delete h; } }

size_t ct_ffn_workspace_bytes(int n_ref, int n_tgt) {
    if (n_ref <= 0 || n_tgt <= 0) return 0;
    return (size_t)2 * (n_ref + n_tgt) * HID * sizeof(float) + 512;     // hidden + projected, both sets
}

static int gemm(const float* A, int lda, const float* B, float* Cm, int M, int N, int K, const float* bn, hipStream_t st) {
    { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, lda, B, Cm, M, N, K, bn); else hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, lda, B, Cm, M, N, K, bn); }
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_ffn_pairgrid(ct_ffn_t* h, const float* feat_ref, int n, const float* feat_tgt, int m, float* corr,
                    void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!h || !feat_ref || !feat_tgt || !corr || !workspace || n <= 0 || m <= 0) return CT_EINVAL;
    DeviceGuard dg(h->device);
    if (dg.err != hipSuccess) return (int)dg.err;
    if (workspace_bytes < ct_ffn_workspace_bytes(n, m)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* Hr = ws; float* Ht = Hr + (size_t)n * HID; float* U = Ht + (size_t)m * HID; float* V = U + (size_t)n * HID;
    int rc;
    (void)rc;
    {   // layer 1 of both point sets, then layer 2 of both (ref half: W2[:512], tgt half: W2[512:]): two launches instead of four
        const int mx = n > m ? n : m;
        { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_pair_kernel<false>), dim3((HID + 63) / 64, (mx + 63) / 64, 2), dim3(256), 0, st,
                           GemmPair{{feat_ref, feat_tgt}, {h->d_w + h->o_w1, h->d_w + h->o_w1}, {Hr, Ht}, {n, m}}, FEAT, HID, FEAT, h->d_w + h->o_bn1); else hipLaunchKernelGGL((gemm_f32_pair_kernel<true>), dim3((HID + 63) / 64, (mx + 63) / 64, 2), dim3(256), 0, st,
                           GemmPair{{feat_ref, feat_tgt}, {h->d_w + h->o_w1, h->d_w + h->o_w1}, {Hr, Ht}, {n, m}}, FEAT, HID, FEAT, h->d_w + h->o_bn1); }
        LAUNCH_CHECK();
        { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_pair_kernel<false>), dim3((HID + 63) / 64, (mx + 63) / 64, 2), dim3(256), 0, st,
                           GemmPair{{Hr, Ht}, {h->d_w + h->o_w2, h->d_w + h->o_w2 + (size_t)HID * HID}, {U, V}, {n, m}}, HID, HID, HID, (const float*)nullptr); else hipLaunchKernelGGL((gemm_f32_pair_kernel<true>), dim3((HID + 63) / 64, (mx + 63) / 64, 2), dim3(256), 0, st,
                           GemmPair{{Hr, Ht}, {h->d_w + h->o_w2, h->d_w + h->o_w2 + (size_t)HID * HID}, {U, V}, {n, m}}, HID, HID, HID, (const float*)nullptr); }
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ffn_pair_kernel, dim3((n + 31) / 32, (m + 31) / 32), dim3(256), 0, st, U, n, V, m,
                       h->d_w + h->o_bn2, h->d_w + h->o_w3, h->b3, corr);
    LAUNCH_CHECK();
    return CT_OK;
}

size_t ct_ffn_predict_workspace_bytes(int rows) { return rows <= 0 ? 0 : (size_t)4 * rows * HID * sizeof(float) + 512; }

int ct_ffn_predict(ct_ffn_t* h, const float* x, int rows, float* out, void* workspace, size_t workspace_bytes,
                   ct_stream_t stream) {
    if (!h || !x || !out || !workspace || rows <= 0) return CT_EINVAL;
    DeviceGuard dg(h->device);
    if (dg.err != hipSuccess) return (int)dg.err;
    if (workspace_bytes < ct_ffn_predict_workspace_bytes(rows)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* H1 = ws; float* H2 = H1 + (size_t)rows * HID; float* U = H2 + (size_t)rows * HID; float* V = U + (size_t)rows * HID;
    int rc;
    if ((rc = gemm(x, 2 * FEAT, h->d_w + h->o_w1, H1, rows, HID, FEAT, h->d_w + h->o_bn1, st))) return rc;
    if ((rc = gemm(x + FEAT, 2 * FEAT, h->d_w + h->o_w1, H2, rows, HID, FEAT, h->d_w + h->o_bn1, st))) return rc;
    if ((rc = gemm(H1, HID, h->d_w + h->o_w2, U, rows, HID, HID, nullptr, st))) return rc;
    if ((rc = gemm(H2, HID, h->d_w + h->o_w2 + (size_t)HID * HID, V, rows, HID, HID, nullptr, st))) return rc;
    hipLaunchKernelGGL(ffn_rows_finish_kernel, dim3(rows), dim3(64), 0, st, U, V, rows, h->d_w + h->o_bn2,
                       h->d_w + h->o_w3, h->b3, out);
    LAUNCH_CHECK();
    return CT_OK;
}

// ------------------------------------------------------------------------------------------------ greedy
size_t ct_greedy_workspace_bytes(int m, int n) {
    if (m <= 0 || n <= 0) return 0;
    const size_t mx = (size_t)(m > n ? m : n);
    return align_up((size_t)m, 256) + align_up((size_t)n, 256) + 3 * align_up((size_t)m * 4, 256) + align_up((size_t)n * 4, 256)
           + align_up(mx * 8, 256) + 256 + 512;
}

int ct_greedy_match(const float* corr, int m, int n, float threshold, int mode, int32_t* pairs, int32_t* n_pairs,
                    double* prior, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!corr || !pairs || !n_pairs || !workspace || m <= 0 || n <= 0 || (mode != 0 && mode != 1)) return CT_EINVAL;
    if ((double)m * (double)n >= 4294967295.0 || (m < n ? m : n) > 16384) return CT_ESHAPE;
    if (workspace_bytes < ct_greedy_workspace_bytes(m, n)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    unsigned char* row_used = ws; ws += align_up((size_t)m, 256);
    unsigned char* col_used = ws; ws += align_up((size_t)n, 256);
    float* rowval = (float*)ws; ws += align_up((size_t)m * 4, 256);
    int* rowcol = (int*)ws; ws += align_up((size_t)m * 4, 256);
    int* row_match = (int*)ws; ws += align_up((size_t)m * 4, 256);
    int* colrow = (int*)ws; ws += align_up((size_t)n * 4, 256);
    unsigned long long* keys = (unsigned long long*)ws; ws += align_up((size_t)(m > n ? m : n) * 8, 256);
    int* ctr = (int*)ws;
    HIPCHK(hipMemsetAsync(row_used, 0, (size_t)(ws - row_used) + 256, st));     // flags, tables and counters
    ENSURE_BIG_LDS(gd_finalize_kernel);
    const int max_rounds = (m < n ? m : n) + 1;
    int hctr[4] = {0, 0, 0, 0};
    // One persistent launch for all rounds (gd_persistent_kernel) when enough of its workgroups are certain to be co-resident: a quarter of the
    // 256-thread slots of the CUs `st` may use (8 per CU), so that up to four such chains on one CU partition cannot block each other.
    // CT_GREEDY_PERSISTENT=0: the launch-per-phase form (A/B; also taken on partitions of fewer than 2 CUs).
    static const bool no_persist = getenv("CT_GREEDY_PERSISTENT") && atoi(getenv("CT_GREEDY_PERSISTENT")) == 0;
    int pg = 0;
    if (!no_persist) {
        int ncu = 0;
        uint32_t mask[16] = {0};
        if (st && hipExtStreamGetCUMask(st, 16, mask) == hipSuccess) { for (int w = 0; w < 16; ++w) ncu += __builtin_popcount(mask[w]); }
        else (void)hipGetLastError();
        if (ncu <= 0) {                                                           // (the null stream / no mask: the whole device)
            int dev = 0; hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        }
        // co-resident 256-thread workgroups per CU: asked from the runtime (register allocation decides: 72 VGPRs -> 7, not 8), a quarter of them
        // per chain so that up to four chains sharing the partition cannot block each other at the device-side barrier
        static const int occ = coresident_per_cu(gd_persistent_kernel, 256, 0);
        const int safe = ncu * occ / 4;
        const int want = (m + 15) / 16 < 64 ? ((m + 15) / 16 < 8 ? 8 : (m + 15) / 16) : 64;
        pg = want < safe ? want : safe;
        if (pg < 4) pg = 0;
    }
    if (pg > 0) {
        hipLaunchKernelGGL(gd_persistent_kernel, dim3(pg), dim3(256), 0, st, corr, m, n, threshold, row_used, col_used, rowval, rowcol, colrow, keys, ctr,
                           max_rounds);
        LAUNCH_CHECK();
        // the pair count stays on the device: gd_finalize_kernel sizes its sort from it, the host only provides LDS for the largest possible count
        int p2 = 1; while (p2 < (m < n ? m : n)) p2 <<= 1;
        hipLaunchKernelGGL(gd_finalize_kernel, dim3(1), dim3(1024), (size_t)p2 * 8, st, keys, ctr, n, pairs, n_pairs);
        LAUNCH_CHECK();
        if (getenv("CT_DEBUG")) {
            HIPCHK(hipMemcpyAsync(hctr, ctr, sizeof(hctr), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            fprintf(stderr, "[ct_greedy_match] m %d n %d pairs %d (persistent, %d workgroups)\n", m, n, hctr[GD_COUNT], pg);
        }
    } else {
    for (int done_rounds = 0; done_rounds < max_rounds;) {
        const int chunk = done_rounds == 0 ? 12 : 8;
        const int nrb = (m + GD_ROWLANES - 1) / GD_ROWLANES, ncb = (n + 63) / 64;
        for (int k = 0; k < chunk; ++k) {
            const int round = done_rounds + k;
            hipLaunchKernelGGL(gd_best_kernel<GD_ROWLANES>, dim3(nrb + ncb), dim3(64 * GD_ROWLANES), 0, st, corr, m, n, row_used, col_used, rowval, rowcol,
                               colrow, ctr, round, nrb);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(gd_accept_kernel, dim3((m + 255) / 256), dim3(256), 0, st, m, n, threshold, rowval, rowcol, colrow,
                               row_used, col_used, keys, ctr, round);
            LAUNCH_CHECK();
        }
        done_rounds += chunk;
        HIPCHK(hipMemcpyAsync(hctr, ctr, sizeof(hctr), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (hctr[GD_DONE] || hctr[GD_NEW + ((done_rounds - 1) & 1)] == 0) break;      // the last round accepted nothing
    }
    if (getenv("CT_DEBUG")) fprintf(stderr, "[ct_greedy_match] m %d n %d pairs %d\n", m, n, hctr[GD_COUNT]);
    {
        int p2 = 1; while (p2 < hctr[GD_COUNT]) p2 <<= 1;
        hipLaunchKernelGGL(gd_finalize_kernel, dim3(1), dim3(1024), (size_t)p2 * 8, st, keys, ctr, n, pairs, n_pairs);
        LAUNCH_CHECK();
    }
    }
    if (prior) {
        hipLaunchKernelGGL(row_match_kernel, dim3((m + 255) / 256), dim3(256), 0, st, row_match, m);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(row_match_set_kernel, dim3((n + 255) / 256), dim3(256), 0, st, row_match, pairs, n_pairs);
        LAUNCH_CHECK();
        const size_t tot = (size_t)m * n;
        hipLaunchKernelGGL(prior_fill_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, prior, m, n, mode,
                           pairs, n_pairs, row_match);
        LAUNCH_CHECK();
    }
    return CT_OK;
}

// ------------------------------------------------------------------------------------------------ PR-GLS
namespace {
struct PrglsWs {
    double *G, *Gln, *M, *P, *part, *dvec, *sqd, *rhs, *C, *predn, *predl, *rowpart, *rowpart0, *normpart, *sc;
    double *U, *Spart, *ypart, *q, *resid, *respart; int* rank;
    double *trb, *tra;                          // P^T Y [n][3] and the row sums of P [m]: sigma2 by the trace identity (scalars_kernel)
};
size_t prgls_layout(int m, int n, int l, unsigned char* base, PrglsWs* w) {
    size_t off = 0;
    auto take = [&](size_t count) { double* p = base ? (double*)(base + off) : nullptr; off += align_up(count * sizeof(double), 256); return p; };
    PrglsWs t{};
    t.G = take((size_t)n * n); t.Gln = take((size_t)n * (l > 0 ? l : 1)); t.M = take((size_t)n * n);
    t.P = take((size_t)m * n); t.part = take((size_t)PART_SEG * 4 * n); t.dvec = take(n); t.sqd = take(n);
    t.rhs = take(3 * (size_t)n); t.C = take(3 * (size_t)n); t.predn = take(3 * (size_t)n);
    t.predl = take(3 * (size_t)(l > 0 ? l : 1)); t.rowpart = take(m); t.rowpart0 = take(m); t.normpart = take(n); t.sc = take(S_NUM);
    t.U = take((size_t)LR_RMAX * n); t.ypart = take((size_t)LR_RMAX * 3);
    t.Spart = take((size_t)LR_RMAX * LR_RMAX > 3 * (size_t)n ? (size_t)LR_RMAX * LR_RMAX : 3 * (size_t)n);   // also the dense path's [3][n] right-hand sides
    t.q = take(LR_RMAX * 3); t.resid = take(n); t.respart = take(2 * (size_t)n); t.rank = (int*)take(8);
    t.trb = take(3 * (size_t)n); t.tra = take(m);
    if (w) *w = t;
    return off;
}

int cholesky_solve(const PrglsWs& w, int n, hipStream_t st) {
    double* W = w.Spart;                       // [3][n]: the low-rank path's S buffer is idle here and sized max(128 * 128, 3 n)
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = n - k0 < NB ? n - k0 : NB;
        const int rem = n - k0 - nb;
        const int nrow = rem > 0 ? (rem + NB - 1) / NB : 0;
        hipLaunchKernelGGL(chol_panel_kernel, dim3(nrow + 1), dim3(256), 0, st, w.M, W, n, k0);
        LAUNCH_CHECK();
        if (rem > 0) {
            hipLaunchKernelGGL(chol_update_kernel, dim3(nrow, nrow + 1), dim3(256), 0, st, w.M, W, n, k0);
            LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(chol_backward_kernel, dim3(1), dim3(256), 0, st, w.M, n, W, w.sqd, w.C);
    LAUNCH_CHECK();
    return CT_OK;
}

// one E-step + M-step solve; leaves C in w.C and sumP in the scalar block.
// rank > 0: low-rank Woodbury M-step (4 launches); rank <= 0: dense blocked Cholesky.
int em_half(const PrglsWs& w, const double* prior, const double* tgt, int m, int n, const double* xref, double lambda,
            int legacy, double vol, int rank, hipStream_t st, bool trace = false, bool want_P = true) {
    // low-rank iterations of the TrackerLite dialect: fused E-step (the posterior is written only if somebody reads it: the caller, or
    // the direct sigma2 sum when the trace identity is switched off)
    int nseg = CS_SEG;
    // three launches per iteration (em_last_block): E-step + finish, Gram + solve + coefficients here, field application + scalars in the caller
    if (rank > 0 && !legacy && vol == 1.0 && trace && em_fuse() && gram_tiled() && xref == w.predn &&
          launch_estep_rows_finish(n, st, prior, w.predn, tgt, m, w.sc, want_P ? w.P : (double*)nullptr, w.part, w.tra,
                                   FinishTail{w.rank + EM_TICKET, w.dvec, w.sqd, w.rhs, w.trb})) {
        LAUNCH_CHECK();
        const int ntile = (rank + LG_T - 1) / LG_T, nwave = ntile * (ntile + 1) / 2 + ntile;
        const size_t lds = ((size_t)rank * (rank + 1) / 2 + 3 * (size_t)rank) * sizeof(double);          // packed triangle + 3 right-hand sides
        hipLaunchKernelGGL(lr_gram_solve_coeff_kernel, dim3((nwave + 3) / 4), dim3(256), lds, st, n, w.U, w.rank, w.sc, w.dvec, w.sqd, w.rhs, w.Spart, w.ypart,
                           SolveTail{w.rank + EM_TICKET + 1, lambda, w.sc, w.q, w.C});
        LAUNCH_CHECK();
        return CT_OK;
    }
    if (rank > 0 && !legacy && vol == 1.0 &&
          launch_estep_cols(n, 1, st, prior, w.predn, n, tgt, m, w.sc, (want_P || !trace) ? w.P : (double*)nullptr, w.part, Bt{0, nullptr},
                            (const double*)nullptr, (const int*)nullptr, 0, trace ? w.tra : (double*)nullptr)) nseg = ES_SEG;
    else {
        hipLaunchKernelGGL(posterior_kernel, dim3((m + 3) / 4), dim3(256), 0, st, prior, w.predn, n, tgt, m, w.sc, legacy, vol, w.P, 0.0, 0.0,
                           Bt{0, nullptr}, (const double*)nullptr, (const int*)nullptr, 0, trace ? w.tra : (double*)nullptr);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(colstats_kernel, dim3((n + 63) / 64, CS_SEG), dim3(256), 0, st, w.P, tgt, m, n, w.part, w.sc);
    }
    LAUNCH_CHECK();
    if (rank > 0) {
        hipLaunchKernelGGL(colstats_finish_par_kernel, dim3((n + 63) / 64), dim3(256), 0, st, w.part, n, xref, w.sc, w.dvec, w.sqd, w.rhs,
                           Bt{0, nullptr}, trace ? w.trb : (double*)nullptr, nseg);
        LAUNCH_CHECK();
        const int nent = rank * (rank + 1) / 2 + 3 * rank;
        const int ntile = (rank + LG_T - 1) / LG_T, nwave = ntile * (ntile + 1) / 2 + ntile;
        if (gram_tiled())
            hipLaunchKernelGGL(lr_gram_tiled_kernel, dim3((nwave + 3) / 4), dim3(256), 0, st, n, w.U, w.rank, w.sc, w.dvec, w.sqd, w.rhs, w.Spart, w.ypart);
        else
            hipLaunchKernelGGL(lr_gram_kernel, dim3((nent + 3) / 4), dim3(256), 0, st, n, w.U, w.rank, w.sc, w.dvec, w.sqd, w.rhs, w.Spart, w.ypart);
        LAUNCH_CHECK();
        const size_t lds = ((size_t)rank * (rank + 1) / 2 + 3 * (size_t)rank) * sizeof(double);          // packed triangle + 3 right-hand sides
        hipLaunchKernelGGL(lr_solve_kernel, dim3(1), dim3(256), lds, st, w.Spart, w.ypart, n, w.rank, lambda, w.dvec, w.sc, w.q);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(lr_coeff_kernel, dim3((n + 63) / 64), dim3(256), 0, st, w.U, n, w.rank, w.q, w.sqd, w.rhs, w.sc, w.C);
        LAUNCH_CHECK();
        return CT_OK;
    }
    if (n <= DS_MAXN) {
        ENSURE_BIG_LDS(dense_small_solve_kernel);
        const size_t lds = (size_t)(n + 3) * (n | 1) * sizeof(double);
        hipLaunchKernelGGL(dense_small_solve_kernel, dim3(1), dim3(256), lds, st, w.part, n, xref, w.G, lambda, w.sc, w.dvec, w.sqd, w.rhs, w.C);
        LAUNCH_CHECK();
        return CT_OK;
    }
    hipLaunchKernelGGL(colstats_finish_kernel, dim3(1), dim3(256), 0, st, w.part, n, xref, lambda, w.sc, w.dvec, w.sqd, w.rhs);
    LAUNCH_CHECK();
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, w.G, w.sqd, w.sc, n, w.M, w.rhs, w.Spart);
    LAUNCH_CHECK();
    return cholesky_solve(w, n, st);
}

// pivoted Cholesky of w.G -> w.U; returns the rank (host value; one stream sync), <= 0 => use the dense path
int lowrank_prepare(const PrglsWs& w, int n, hipStream_t st, int* rank_coarse, int* rank_fine) {
    ENSURE_BIG_LDS(lr_solve_kernel);
    ENSURE_LDS(lr_gram_solve_coeff_kernel, (LR_RMAX * (LR_RMAX + 1) / 2 + 3 * LR_RMAX) * 8);
    ENSURE_LDS(em_persistent_kernel, (LR_RMAX * (LR_RMAX + 1) / 2 + 3 * LR_RMAX) * 8);
    hipLaunchKernelGGL(lowrank_factor_kernel, dim3(1), dim3(1024), 0, st, w.G, n, kLowRankTol, kLowRankTolTight, w.U, w.resid, w.rank);
    LAUNCH_CHECK();
    int r[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(r, w.rank, sizeof(r), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *rank_coarse = r[0]; *rank_fine = r[1];
    return CT_OK;
}

// Batched initialisation (one launch for all problems): G = gauss(ref, ref), Gln = gauss(ref, tracked) stored [l][n], T(X) = X,
// tracked copy, scalar block {sigma2 0, gamma 0.05, ...} -- the arithmetic of gauss_kernel / the host-side init of
// ct_prgls_two_ref (trackerlite.py:319-325).
__global__ __launch_bounds__(256) void prgls_init_batch_kernel(const double* __restrict__ ref, const double* __restrict__ tracked,
                                                               double two_b2, double* __restrict__ G, double* __restrict__ Gln,
                                                               double* __restrict__ predn, double* __restrict__ predl,
                                                               double* __restrict__ sc, double* __restrict__ csum, Bt bt) {
    BT_SHIFT(const double*, ref); BT_SHIFT(const double*, tracked); BT_SHIFT(double*, G); BT_SHIFT(double*, Gln);
    BT_SHIFT(double*, predn); BT_SHIFT(double*, predl); BT_SHIFT(double*, sc); BT_SHIFT(double*, csum);
    const int n = bt.dims[4 * blockIdx.z + 1], l = bt.dims[4 * blockIdx.z + 2];
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid < (size_t)n * n) {
        const int i = (int)(gid / n), j = (int)(gid - (size_t)i * n);
        const double dx = ref[3 * j] - ref[3 * i], dy = ref[3 * j + 1] - ref[3 * i + 1], dz = ref[3 * j + 2] - ref[3 * i + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        G[gid] = exp(-d2 / two_b2);
    }
    if (gid < (size_t)n * l) {
        const int i = (int)(gid / n), j = (int)(gid - (size_t)i * n);
        const double dx = ref[3 * j] - tracked[3 * i], dy = ref[3 * j + 1] - tracked[3 * i + 1], dz = ref[3 * j + 2] - tracked[3 * i + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        Gln[gid] = exp(-d2 / two_b2);
    }
    if (gid < 3 * (size_t)n) { predn[gid] = ref[gid]; csum[gid] = 0.0; }
    if (gid < 3 * (size_t)l) predl[gid] = tracked[gid];
    if (gid < S_NUM) sc[gid] = gid == S_GAMMA ? 0.05 : 0.0;
}
__global__ void prgls_stop_kernel(double* __restrict__ sc, size_t stride, int b) { sc[(size_t)b * stride + S_DONE] = 1.0; }
// scalar blocks / ranks of all problems -> one contiguous buffer (a strided hipMemcpy2DAsync to the host was observed to
// serialise with every other stream of the process: 15 instead of 65 volumes/s in the pipelined benchmark)
__global__ void prgls_gather_kernel(const double* __restrict__ sc, const int* __restrict__ rank, size_t stride, int B,
                                    double* __restrict__ out_sc, int* __restrict__ out_rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (out_sc && i < B * S_NUM) out_sc[i] = sc[(size_t)(i / S_NUM) * stride + i % S_NUM];
    if (out_rank && i < 2 * B) out_rank[i] = reinterpret_cast<const int*>(reinterpret_cast<const double*>(rank) + (size_t)(i / 2) * stride)[i % 2];
}

struct BatchLayout { size_t per_bytes, in_prior, in_tgt, in_ref, in_trk, dims_off, tail_off, total; };
BatchLayout batch_layout(int B, int mm, int nn, int ll) {
    BatchLayout L{};
    size_t off = prgls_layout(mm, nn, ll, nullptr, nullptr);
    L.in_prior = off; off += align_up((size_t)mm * nn * sizeof(double), 256);
    L.in_tgt = off; off += align_up(3 * (size_t)mm * sizeof(double), 256);
    L.in_ref = off; off += align_up(3 * (size_t)nn * sizeof(double), 256);
    L.in_trk = off; off += align_up(3 * (size_t)(ll > 0 ? ll : 1) * sizeof(double), 256);
    L.per_bytes = off;
    L.dims_off = (size_t)B * L.per_bytes;
    L.tail_off = L.dims_off + align_up((size_t)B * 4 * sizeof(int), 256) + align_up((size_t)B * (S_NUM * sizeof(double) + 2 * sizeof(int)), 256);
    L.total = L.tail_off + ct_prgls_workspace_bytes(mm, nn, ll) + 512;
    return L;
}
}  // namespace

size_t ct_prgls_workspace_bytes(int m, int n, int l) {
    if (m <= 0 || n <= 0 || l < 0) return 0;
    // EM state + (legacy dialect) the prior built inside, its pair list and the greedy scratch
    return prgls_layout(m, n, l, nullptr, nullptr) + align_up((size_t)m * n * sizeof(double), 256)
           + align_up((size_t)n * 8 + 4, 256) + ct_greedy_workspace_bytes(m, n) + 1024;
}

// What ct_prgls_two_ref needs of the reference set alone: Gram matrix, its pivoted-Cholesky factor, the two ranks (ct_prgls_prepare_ref)
namespace {
constexpr unsigned long long PREP_MAGIC = 0x3153474c52505443ull;      // "CTPRGLS1"
struct PreparedHdr { unsigned long long magic; int n; int pad; double beta; int rank[2]; };
struct PreparedRef { PreparedHdr* hdr; double* G; double* U; double* resid; };
// the header travels as a kernel argument: a hipMemcpyAsync from a stack struct is only safe because pageable copies are staged synchronously
__global__ void prepared_hdr_kernel(PreparedHdr* dst, const PreparedHdr h) { *dst = h; }
size_t prepared_layout(int n, unsigned char* base, PreparedRef* p) {
    size_t off = 256;
    auto take = [&](size_t count) { double* q = base ? (double*)(base + off) : nullptr; off += align_up(count * sizeof(double), 256); return q; };
    PreparedRef t{};
    t.hdr = (PreparedHdr*)base; t.G = take((size_t)n * n); t.U = take((size_t)LR_RMAX * n); t.resid = take(n);
    if (p) *p = t;
    return off;
}

int prgls_two_ref_impl(const double* prior, const double* tgt, int m, const double* ref, int n, const double* tracked, int l,
                       double beta, double lambda, int max_iteration, double* out_tracked, double* out_ref, double* posterior,
                       int* iters, void* workspace, size_t workspace_bytes, const void* prepared, size_t prepared_bytes, ct_stream_t stream) {
    if (!prior || !tgt || !ref || !workspace || m <= 0 || n <= 0 || l < 0 || (l > 0 && (!tracked || !out_tracked))) return CT_EINVAL;
    if (workspace_bytes < ct_prgls_workspace_bytes(m, n, l)) return CT_EWORKSPACE;
    if (prepared && (((uintptr_t)prepared & 255) || prepared_bytes < prepared_layout(n, nullptr, nullptr))) return CT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    PrglsWs w;
    prgls_layout(m, n, l, (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), &w);
    HIPCHK(hipMemsetAsync(w.rank + EM_TICKET, 0, 3 * sizeof(int), (hipStream_t)stream));     // tickets of the fused EM kernels (em_last_block)
    PreparedRef prep{};
    const size_t nn = (size_t)n * n;
    // init (trackerlite.py:319-325): gamma 0.05, Gram matrices with beta^2, sigma2 = mean d2 / 3, T(X) = X
    if (prepared) {                                           // the kernels below only read G, U: they may live in the caller's buffer
        prepared_layout(n, (unsigned char*)prepared, &prep);
        w.G = prep.G; w.U = prep.U;
        HIPCHK(hipMemcpyAsync(w.rank, prep.hdr->rank, 2 * sizeof(int), hipMemcpyDeviceToDevice, st));     // (the residual monitor raises ITS copy)
    } else {
        hipLaunchKernelGGL(gauss_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, ref, n, ref, n, 2.0 * beta * beta, w.G, 0);
        LAUNCH_CHECK();
    }
    if (l > 0) {   // tracked-set kernel stored transposed [l][n] so that the field application reads rows
        const size_t nl = (size_t)n * l;
        hipLaunchKernelGGL(gauss_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, ref, n, tracked, l, 2.0 * beta * beta, w.Gln, 0);
        LAUNCH_CHECK();
        HIPCHK(hipMemcpyAsync(w.predl, tracked, 3 * (size_t)l * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipMemcpyAsync(w.predn, ref, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    double init_sc[S_NUM] = {0.0, 0.05, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    HIPCHK(hipMemcpyAsync(w.sc, init_sc, sizeof(init_sc), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((m + 3) / 4), dim3(256), 0, st, ref, n, tgt, m, (const double*)nullptr, w.rowpart0,
                       (const double*)nullptr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(256), 0, st, w.rowpart0, m, n, 0, w.sc, (const double*)nullptr, (const double*)nullptr);
    LAUNCH_CHECK();
    // `rank` sizes the launches (grids, LDS) for the finest rank; the kernels read the rank in force from w.rank[0], which
    // starts at the coarse rank and is raised on the device by the residual monitor (scalars_kernel)
    int rank = 0, rank_coarse = 0, rc;
    if (prepared) {                                           // the ranks (and the tag that says whose they are) come from the prepared header
        PreparedHdr h{};
        HIPCHK(hipMemcpyAsync(&h, prep.hdr, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h.magic != PREP_MAGIC || h.n != n || h.beta != beta) return CT_EINVAL;     // not what ct_prgls_prepare_ref wrote for this n / beta
        ENSURE_BIG_LDS(lr_solve_kernel);
        ENSURE_LDS(lr_gram_solve_coeff_kernel, (LR_RMAX * (LR_RMAX + 1) / 2 + 3 * LR_RMAX) * 8);
        ENSURE_LDS(em_persistent_kernel, (LR_RMAX * (LR_RMAX + 1) / 2 + 3 * LR_RMAX) * 8);
    ENSURE_LDS(em_persistent_kernel, (LR_RMAX * (LR_RMAX + 1) / 2 + 3 * LR_RMAX) * 8);
        rank_coarse = h.rank[0]; rank = h.rank[1];
    } else if ((rc = lowrank_prepare(w, n, st, &rank_coarse, &rank))) return rc;
    if (rank_coarse <= 0 || getenv("CT_PRGLS_DENSE")) rank = 0;
    // EM iterations are enqueued in chunks; every kernel returns immediately once the device-side
    // convergence flag is set, the host looks at the flag once per chunk (trackerlite.py:353-356).
    // The low-rank M-step is verified on the fly against the exact Gram matrix (S_RES); if its residual
    // is not negligible the whole loop is redone with the dense Cholesky M-step.
    const int total = max_iteration - 1;
    int done_iters = 0, lr_level = 0;
    double hsc[S_NUM] = {0};
    // state checkpoint (ref set, tracked set, scalars) taken at every chunk start: if the monitor rejects the
    // low-rank M-step inside a chunk, that chunk is redone with the dense M-step from the last verified state
    double* ck_n = w.M;                                   // the dense-path matrix is idle while the low-rank path runs
    double* ck_l = w.M + 3 * (size_t)n;
    double* ck_sc = w.M + 3 * (size_t)(n + (l > 0 ? l : 1));
    const bool ck_fits = (size_t)3 * (n + (l > 0 ? l : 1)) + S_NUM <= (size_t)n * n;
    for (int enq = 0; enq < total;) {
        const int chunk = prgls_chunk(enq, total);
        if (rank > 0 && ck_fits) {
            HIPCHK(hipMemcpyAsync(ck_n, w.predn, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
            if (l > 0) HIPCHK(hipMemcpyAsync(ck_l, w.predl, 3 * (size_t)l * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(ck_sc, w.sc, S_NUM * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        // The chunk as ONE persistent launch (em_persistent_kernel) where its grid is certain to be co-resident several times over on the CUs
        // `st` may use (as for the greedy: a quarter of their 256-thread slots) and the shape is the one its E-step is built for.
        // OFF by default (CT_EM_PERSISTENT=1 enables it) -- built in round 5, bit-identical, measured slower: six device-side barriers and the
        // replicated solve make an iteration 112 us alone against 72 for the seven launches, and beside the U-Net the spinning workgroups cost
        // the conv stream more than the launches they replace (frame loop 6.86 against 6.47 ms per frame; profiles/r05_conv_experiments.txt).
        static const bool no_emp = !(getenv("CT_EM_PERSISTENT") && atoi(getenv("CT_EM_PERSISTENT")) == 1);
        if (rank > 0 && sigma_trace() && !no_emp && !em_fuse() && gram_tiled() && estep_mode() >= 2 && estep_nq() == EMP_NQ && n <= 64 * EMP_NQ * EMP_W) {
            int ncu = 0;
            uint32_t mask[16] = {0};
            if (st && hipExtStreamGetCUMask(st, 16, mask) == hipSuccess) { for (int q = 0; q < 16; ++q) ncu += __builtin_popcount(mask[q]); }
            else (void)hipGetLastError();
            if (ncu <= 0) {
                int dev = 0; hipDeviceProp_t pr;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
            }
            const size_t lds_need = ((size_t)rank * (rank + 1) / 2 + 3 * (size_t)rank) * sizeof(double);
            int pg = ncu * coresident_per_cu(em_persistent_kernel, 256, lds_need) / 4;      // (a quarter of what the runtime says fits, LDS included)
            if (pg > ncu / 2) pg = ncu / 2;
            if (pg > 64) pg = 64;
            if (pg >= 8) {
                int* bar = w.rank + EM_TICKET + 3;
                HIPCHK(hipMemsetAsync(bar, 0, sizeof(int), st));
                const size_t lds = ((size_t)rank * (rank + 1) / 2 + 3 * (size_t)rank) * sizeof(double);
                const EmP ea{prior, tgt, m, n, l, lambda, w.predn, w.predl, w.G, w.Gln, posterior != nullptr ? w.P : (double*)nullptr, w.part, w.dvec, w.sqd,
                             w.rhs, w.trb, w.tra, w.U, w.rank, w.Spart, w.ypart, w.q, w.C, w.normpart, w.respart, w.rowpart, w.sc, bar, chunk};
                hipLaunchKernelGGL(em_persistent_kernel, dim3(pg), dim3(256), lds, st, ea);
                LAUNCH_CHECK();
                goto chunk_done;
            }
        }
        for (int k = 0; k < chunk; ++k) {
            // low-rank iterations take sigma2 from the trace identity (as the batched chain does: the two stay bit-identical);
            // the dense continuation keeps the direct sum
            const bool trace = rank > 0 && sigma_trace();
            if ((rc = em_half(w, prior, tgt, m, n, w.predn, lambda, 0, 1.0, rank, st, trace, posterior != nullptr))) return rc;
            if (trace && em_fuse()) {                             // field application with the iteration's scalars as its tail
                hipLaunchKernelGGL(apply_dual_scalars_kernel, dim3((n + l + 3) / 4), dim3(256), 0, st, w.C, w.G, n, w.predn, w.Gln, l, w.predl,
                                   w.normpart, w.sc, w.dvec, w.sqd, w.rhs, w.respart,
                                   ScalarsTail{w.rank + EM_TICKET + 2, w.sc, m, 1, w.rank, tgt, w.tra, w.trb, w.rowpart});
                LAUNCH_CHECK();
                continue;
            }
            hipLaunchKernelGGL(apply_dual_kernel, dim3((n + l + 3) / 4), dim3(256), 0, st, w.C, w.G, n, w.predn, w.Gln, l, w.predl,
                               w.normpart, w.sc, w.dvec, w.sqd, w.rhs, rank > 0 ? w.respart : (double*)nullptr);
            LAUNCH_CHECK();
            if (trace) {
                hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(256), 0, st, w.rowpart, m, n, 1, w.sc, w.normpart, w.respart, w.rank, Bt{0, nullptr},
                                   tgt, (const double*)w.tra, (const double*)w.predn, (const double*)w.dvec, (const double*)w.trb);
            } else {
                hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((m + 3) / 4), dim3(256), 0, st, w.predn, n, tgt, m, w.P, w.rowpart, w.sc);
                LAUNCH_CHECK();
                hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(256), 0, st, w.rowpart, m, n, 1, w.sc, w.normpart,
                                   rank > 0 ? w.respart : (const double*)nullptr, rank > 0 ? w.rank : (int*)nullptr);
            }
            LAUNCH_CHECK();
        }
    chunk_done:
        HIPCHK(hipMemcpyAsync(hsc, w.sc, sizeof(hsc), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (rank > 0 && !(hsc[S_RES] <= kLowRankMaxResidual)) {
            // first rejection: redo the chunk from its checkpoint with the fine rank in force from its first iteration (the
            // device had raised it at most part of the way through); second rejection: dense path
            const bool retry_fine = lr_level == 0 && ck_fits;
            lr_level = 1;
            if (getenv("CT_DEBUG"))
                fprintf(stderr, "[ct_prgls_two_ref] low-rank M-step rejected after %d iterations (ranks %d/%d, residual %.3e, sigma2 %.3e): %s from the last checkpoint\n",
                        (int)hsc[S_IT], rank_coarse, rank, hsc[S_RES], hsc[S_SIGMA2], retry_fine ? "fine rank throughout" : "dense");
            if (retry_fine) {
                HIPCHK(hipMemcpyAsync(w.rank, w.rank + 1, sizeof(int), hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(w.predn, ck_n, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (l > 0) HIPCHK(hipMemcpyAsync(w.predl, ck_l, 3 * (size_t)l * sizeof(double), hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(w.sc, ck_sc, S_NUM * sizeof(double), hipMemcpyDeviceToDevice, st));
                continue;
            }
            rank = 0;
            if (ck_fits) {           // restore the last verified state and redo this chunk
                HIPCHK(hipMemcpyAsync(w.predn, ck_n, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (l > 0) HIPCHK(hipMemcpyAsync(w.predl, ck_l, 3 * (size_t)l * sizeof(double), hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(w.sc, ck_sc, S_NUM * sizeof(double), hipMemcpyDeviceToDevice, st));
            } else {                 // no room for a checkpoint: restart from the initial state
                HIPCHK(hipMemcpyAsync(w.predn, ref, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (l > 0) HIPCHK(hipMemcpyAsync(w.predl, tracked, 3 * (size_t)l * sizeof(double), hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(w.sc, init_sc, sizeof(init_sc), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(256), 0, st, w.rowpart0, m, n, 0, w.sc, (const double*)nullptr, (const double*)nullptr);
                LAUNCH_CHECK();
                enq = 0;
            }
            continue;
        }
        enq += chunk;
        done_iters = (int)hsc[S_IT];
        if (hsc[S_DONE] != 0.0) break;
    }
    if (getenv("CT_DEBUG"))
        fprintf(stderr, "[ct_prgls_two_ref] ranks %d/%d%s iterations %d done %g residual %.3e sigma2 %.6e norm2 %.3e\n",
                rank_coarse, rank, rank > 0 ? "" : " (dense)", done_iters, hsc[S_DONE], hsc[S_RES], hsc[S_SIGMA2], hsc[S_NORM2]);
    if (iters) *iters = done_iters;
    if (l > 0) HIPCHK(hipMemcpyAsync(out_tracked, w.predl, 3 * (size_t)l * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (out_ref) HIPCHK(hipMemcpyAsync(out_ref, w.predn, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (posterior) HIPCHK(hipMemcpyAsync(posterior, w.P, (size_t)m * n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return CT_OK;
}
}  // namespace

int ct_prgls_two_ref(const double* prior, const double* tgt, int m, const double* ref, int n, const double* tracked, int l,
                     double beta, double lambda, int max_iteration, double* out_tracked, double* out_ref, double* posterior,
                     int* iters, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    return prgls_two_ref_impl(prior, tgt, m, ref, n, tracked, l, beta, lambda, max_iteration, out_tracked, out_ref, posterior, iters, workspace,
                              workspace_bytes, nullptr, 0, stream);
}

size_t ct_prgls_prepared_bytes(int n) { return n > 0 ? prepared_layout(n, nullptr, nullptr) : 0; }

int ct_prgls_prepare_ref(const double* ref, int n, double beta, void* prepared, size_t prepared_bytes, ct_stream_t stream) {
    if (!ref || !prepared || n <= 0 || ((uintptr_t)prepared & 255)) return CT_EINVAL;
    if (prepared_bytes < prepared_layout(n, nullptr, nullptr)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    PreparedRef p;
    prepared_layout(n, (unsigned char*)prepared, &p);
    const PreparedHdr h{PREP_MAGIC, n, 0, beta, {0, 0}};
    hipLaunchKernelGGL(prepared_hdr_kernel, dim3(1), dim3(1), 0, st, p.hdr, h);
    LAUNCH_CHECK();
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(gauss_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, ref, n, ref, n, 2.0 * beta * beta, p.G, 0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(lowrank_factor_kernel, dim3(1), dim3(1024), 0, st, p.G, n, kLowRankTol, kLowRankTolTight, p.U, p.resid, p.hdr->rank);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_prgls_two_ref_prepared(const double* prior, const double* tgt, int m, const double* ref, int n, const double* tracked, int l,
                              double beta, double lambda, int max_iteration, double* out_tracked, double* out_ref, double* posterior,
                              int* iters, void* workspace, size_t workspace_bytes, const void* prepared, size_t prepared_bytes,
                              ct_stream_t stream) {
    if (!prepared) return CT_EINVAL;
    return prgls_two_ref_impl(prior, tgt, m, ref, n, tracked, l, beta, lambda, max_iteration, out_tracked, out_ref, posterior, iters, workspace,
                              workspace_bytes, prepared, prepared_bytes, stream);
}

size_t ct_prgls_batched_workspace_bytes(int B, const int* m, const int* n, const int* l) {
    if (B <= 0 || !m || !n || !l) return 0;
    int mm = 0, nn = 0, ll = 0;
    for (int b = 0; b < B; ++b) {
        if (m[b] <= 0 || n[b] <= 0 || l[b] < 0) return 0;
        mm = m[b] > mm ? m[b] : mm; nn = n[b] > nn ? n[b] : nn; ll = l[b] > ll ? l[b] : ll;
    }
    return batch_layout(B, mm, nn, ll).total;
}

int ct_prgls_two_ref_batched(int B, const double* const* prior, const double* const* tgt, const int* m, const double* const* ref,
                             const int* n, const double* const* tracked, const int* l, double beta, double lambda, int max_iteration,
                             double* const* out_tracked, double* const* out_ref, double* const* posterior, int* iters,
                             void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (B <= 0 || !prior || !tgt || !m || !ref || !n || !l || !workspace) return CT_EINVAL;
    int mm = 0, nn = 0, ll = 0;
    for (int b = 0; b < B; ++b) {
        if (!prior[b] || !tgt[b] || !ref[b] || m[b] <= 0 || n[b] <= 0 || l[b] < 0 ||
            (l[b] > 0 && (!tracked || !tracked[b] || !out_tracked || !out_tracked[b]))) return CT_EINVAL;
        mm = m[b] > mm ? m[b] : mm; nn = n[b] > nn ? n[b] : nn; ll = l[b] > ll ? l[b] : ll;
    }
    const BatchLayout L = batch_layout(B, mm, nn, ll);
    if (workspace_bytes < L.total) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* base = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    PrglsWs w;
    prgls_layout(mm, nn, ll, base, &w);                      // problem 0's arrays; problem b's are `stride` doubles further
    const size_t stride = L.per_bytes / sizeof(double);
    double* in_prior = (double*)(base + L.in_prior); double* in_tgt = (double*)(base + L.in_tgt);
    double* in_ref = (double*)(base + L.in_ref); double* in_trk = (double*)(base + L.in_trk);
    int* d_dims = (int*)(base + L.dims_off);
    double* d_gsc = (double*)(base + L.dims_off + align_up((size_t)B * 4 * sizeof(int), 256));      // gathered scalar blocks
    int* d_grank = (int*)(d_gsc + (size_t)B * S_NUM);                                                // gathered ranks
    std::vector<int> hdims(4 * (size_t)B, 0);
    for (int b = 0; b < B; ++b) {
        hdims[4 * b] = m[b]; hdims[4 * b + 1] = n[b]; hdims[4 * b + 2] = l[b];
        HIPCHK(hipMemcpyAsync(in_prior + b * stride, prior[b], (size_t)m[b] * n[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(in_tgt + b * stride, tgt[b], 3 * (size_t)m[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(in_ref + b * stride, ref[b], 3 * (size_t)n[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (l[b] > 0) HIPCHK(hipMemcpyAsync(in_trk + b * stride, tracked[b], 3 * (size_t)l[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipMemcpyAsync(d_dims, hdims.data(), hdims.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                        // hdims is a host temporary
    const Bt bt{stride, d_dims};
    const unsigned zB = (unsigned)B;
    {
        const size_t tot = (size_t)nn * (nn > ll ? nn : ll);
        const size_t cover = tot > 3 * (size_t)(nn > ll ? nn : ll) ? tot : 3 * (size_t)(nn > ll ? nn : ll);
        hipLaunchKernelGGL(prgls_init_batch_kernel, dim3((unsigned)((cover + 255) / 256), 1, zB), dim3(256), 0, st, in_ref, in_trk,
                           2.0 * beta * beta, w.G, w.Gln, w.predn, w.predl, w.sc, w.M, bt);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((mm + 3) / 4, 1, zB), dim3(256), 0, st, in_ref, nn, in_tgt, mm, (const double*)nullptr,
                       w.rowpart0, (const double*)nullptr, bt);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(scalars_kernel, dim3(1, 1, zB), dim3(256), 0, st, w.rowpart0, mm, nn, 0, w.sc, (const double*)nullptr,
                       (const double*)nullptr, (int*)nullptr, bt);
    LAUNCH_CHECK();
    ENSURE_BIG_LDS(lr_solve_kernel);
    ENSURE_LDS(lr_gram_solve_coeff_kernel, (LR_RMAX * (LR_RMAX + 1) / 2 + 3 * LR_RMAX) * 8);
    ENSURE_LDS(em_persistent_kernel, (LR_RMAX * (LR_RMAX + 1) / 2 + 3 * LR_RMAX) * 8);
    hipLaunchKernelGGL(lowrank_factor_kernel, dim3(1, 1, zB), dim3(1024), 0, st, w.G, nn, kLowRankTol, kLowRankTolTight, w.U, w.resid, w.rank, bt);
    LAUNCH_CHECK();
    std::vector<int> hrank(2 * (size_t)B, 0);
    hipLaunchKernelGGL(prgls_gather_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, st, (const double*)nullptr, w.rank, stride, B, (double*)nullptr, d_grank);
    LAUNCH_CHECK();
    HIPCHK(hipMemcpyAsync(hrank.data(), d_grank, 2 * (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // problems whose Gram matrix does not factorise to the coarse tolerance (or CT_PRGLS_DENSE) and problems whose low-rank
    // M-step the residual monitor rejects are finished by the single-problem routine (its retry / dense logic) afterwards
    std::vector<char> fallback(B, 0), finished(B, 0);
    int rank = 0, live = 0;
    for (int b = 0; b < B; ++b) {
        if (hrank[2 * b] <= 0 || getenv("CT_PRGLS_DENSE")) {
            fallback[b] = 1;
            hipLaunchKernelGGL(prgls_stop_kernel, dim3(1), dim3(1), 0, st, w.sc, stride, b);
            LAUNCH_CHECK();
        } else { rank = hrank[2 * b + 1] > rank ? hrank[2 * b + 1] : rank; ++live; }
    }
    const int total = max_iteration - 1;
    const bool rgm = B >= RG_MIN_BATCH && row_groups();
    const bool defer = ll > 0 && defer_tracked();         // tracked set moved once, after the loop (apply_tracked_kernel); Csum lives in w.M
    const int lla = defer ? 0 : ll;
    // structured priors (prior_scan_kernel): the table follows the coefficient sums in the dense path's matrix, idle in this loop;
    // d_grank has been read back
    const bool sp_on = prior_scan() && 3 * (size_t)nn + 3 * (size_t)mm + 8 <= (size_t)nn * nn;
    double* sp_tab = w.M + 3 * (size_t)nn;
    const bool trace = sigma_trace();                     // sigma2 by the trace identity (scalars_kernel)
    double* tr_b = w.trb; double* tr_a = w.tra;
    int* d_dense = d_grank;
    if (sp_on) {
        HIPCHK(hipMemsetAsync(d_dense, 0, (size_t)B * sizeof(int), st));
        hipLaunchKernelGGL(prior_scan_kernel, dim3((mm + 3) / 4, 1, zB), dim3(256), 0, st, in_prior, mm, nn, sp_tab, mm, d_dense, bt);
        LAUNCH_CHECK();
    }
    std::vector<double> hsc((size_t)B * S_NUM, 0.0);
    bool any_posterior = false;
    for (int b = 0; b < B && posterior; ++b) any_posterior = any_posterior || posterior[b] != nullptr;
    for (int enq = 0; enq < total && live > 0;) {
        const int chunk = prgls_chunk(enq, total);
        for (int k = 0; k < chunk; ++k) {
            int nseg = ES_SEG;
            if (!launch_estep_cols(nn, zB, st, in_prior, w.predn, nn, in_tgt, mm, w.sc, (any_posterior || !trace) ? w.P : (double*)nullptr, w.part, bt,
                                   sp_on ? (const double*)sp_tab : (const double*)nullptr, (const int*)d_dense, mm, trace ? tr_a : (double*)nullptr)) {
                nseg = CS_SEG;
                hipLaunchKernelGGL(posterior_kernel, dim3((mm + 3) / 4, 1, zB), dim3(256), 0, st, in_prior, w.predn, nn, in_tgt, mm, w.sc, 0, 1.0,
                                   w.P, 0.0, 0.0, bt, sp_on ? (const double*)sp_tab : (const double*)nullptr, (const int*)d_dense, mm,
                                   trace ? tr_a : (double*)nullptr);
                LAUNCH_CHECK();
                hipLaunchKernelGGL(colstats_kernel, dim3((nn + 63) / 64, CS_SEG, zB), dim3(256), 0, st, w.P, in_tgt, mm, nn, w.part, w.sc, bt);
            }
            LAUNCH_CHECK();
            hipLaunchKernelGGL(colstats_finish_par_kernel, dim3((nn + 63) / 64, 1, zB), dim3(256), 0, st, w.part, nn, w.predn, w.sc, w.dvec,
                               w.sqd, w.rhs, bt, trace ? tr_b : (double*)nullptr, nseg);
            LAUNCH_CHECK();
            const int nent = rank * (rank + 1) / 2 + 3 * rank;
            const int ntile = (rank + LG_T - 1) / LG_T, nwave = ntile * (ntile + 1) / 2 + ntile;
            if (gram_tiled())
                hipLaunchKernelGGL(lr_gram_tiled_kernel, dim3((nwave + 3) / 4, 1, zB), dim3(256), 0, st, nn, w.U, w.rank, w.sc, w.dvec, w.sqd, w.rhs,
                                   w.Spart, w.ypart, bt);
            else
                hipLaunchKernelGGL(lr_gram_kernel, dim3((nent + 3) / 4, 1, zB), dim3(256), 0, st, nn, w.U, w.rank, w.sc, w.dvec, w.sqd, w.rhs,
                               w.Spart, w.ypart, bt);
            LAUNCH_CHECK();
            const size_t lds = ((size_t)rank * (rank + 1) / 2 + 3 * (size_t)rank) * sizeof(double);
            hipLaunchKernelGGL(lr_solve_kernel, dim3(1, 1, zB), dim3(256), lds, st, w.Spart, w.ypart, nn, w.rank, lambda, w.dvec, w.sc, w.q, bt);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(lr_coeff_kernel, dim3((nn + 63) / 64, 1, zB), dim3(256), 0, st, w.U, nn, w.rank, w.q, w.sqd, w.rhs, w.sc, w.C, bt,
                               defer ? w.M : (double*)nullptr);
            LAUNCH_CHECK();
            if (rgm)
                hipLaunchKernelGGL(apply_dual_rg_kernel, dim3(((nn + RG - 1) / RG + (lla + RG - 1) / RG + 3) / 4, 1, zB), dim3(256), 0, st, w.C, w.G, nn,
                                   w.predn, w.Gln, ll, w.predl, w.normpart, w.sc, w.dvec, w.sqd, w.rhs, w.respart, bt, defer ? 1 : 0);
            else
                hipLaunchKernelGGL(apply_dual_kernel, dim3((nn + lla + 3) / 4, 1, zB), dim3(256), 0, st, w.C, w.G, nn, w.predn, w.Gln, ll, w.predl,
                                   w.normpart, w.sc, w.dvec, w.sqd, w.rhs, w.respart, bt, defer ? 1 : 0);
            LAUNCH_CHECK();
            if (!trace) {
                hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((mm + 3) / 4, 1, zB), dim3(256), 0, st, w.predn, nn, in_tgt, mm, w.P, w.rowpart, w.sc, bt);
                LAUNCH_CHECK();
            }
            if (trace)
                hipLaunchKernelGGL(scalars_kernel, dim3(1, 1, zB), dim3(256), 0, st, w.rowpart, mm, nn, 1, w.sc, w.normpart, w.respart, w.rank, bt,
                                   (const double*)in_tgt, (const double*)tr_a, (const double*)w.predn, (const double*)w.dvec, (const double*)tr_b);
            else
                hipLaunchKernelGGL(scalars_kernel, dim3(1, 1, zB), dim3(256), 0, st, w.rowpart, mm, nn, 1, w.sc, w.normpart, w.respart, w.rank, bt);
            LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(prgls_gather_kernel, dim3((B * S_NUM + 63) / 64), dim3(64), 0, st, w.sc, (const int*)nullptr, stride, B, d_gsc, (int*)nullptr);
        LAUNCH_CHECK();
        HIPCHK(hipMemcpyAsync(hsc.data(), d_gsc, (size_t)B * S_NUM * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        enq += chunk;
        live = 0;
        for (int b = 0; b < B; ++b) {
            if (fallback[b] || finished[b]) continue;
            const double* h = &hsc[(size_t)b * S_NUM];
            if (!(h[S_RES] <= kLowRankMaxResidual)) {
                fallback[b] = 1;
                hipLaunchKernelGGL(prgls_stop_kernel, dim3(1), dim3(1), 0, st, w.sc, stride, b);
                LAUNCH_CHECK();
                continue;
            }
            if (iters) iters[b] = (int)h[S_IT];
            if (h[S_DONE] != 0.0) finished[b] = 1; else ++live;
        }
    }
    if (defer) {
        hipLaunchKernelGGL(apply_tracked_kernel, dim3((ll + 3) / 4, 1, zB), dim3(256), 0, st, w.M, w.Gln, nn, w.predl, ll, (const unsigned char*)nullptr, bt);
        LAUNCH_CHECK();
    }
    for (int b = 0; b < B; ++b) {
        if (fallback[b]) {
            int it = 0;
            const int rc = ct_prgls_two_ref(prior[b], tgt[b], m[b], ref[b], n[b], l[b] > 0 ? tracked[b] : nullptr, l[b], beta, lambda, max_iteration,
                                            l[b] > 0 ? out_tracked[b] : nullptr, out_ref ? out_ref[b] : nullptr,
                                            posterior ? posterior[b] : nullptr, &it, base + L.tail_off, ct_prgls_workspace_bytes(mm, nn, ll) + 256,
                                            stream);
            if (rc) return rc;
            if (iters) iters[b] = it;
            continue;
        }
        if (l[b] > 0) HIPCHK(hipMemcpyAsync(out_tracked[b], w.predl + b * stride, 3 * (size_t)l[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (out_ref && out_ref[b]) HIPCHK(hipMemcpyAsync(out_ref[b], w.predn + b * stride, 3 * (size_t)n[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (posterior && posterior[b]) HIPCHK(hipMemcpyAsync(posterior[b], w.P + b * stride, (size_t)m[b] * n[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    return CT_OK;
}

int ct_prgls_legacy(const double* X, int n, const double* Y, int m, const float* corr, double BETA, int max_iteration,
                    double LAMBDA, double vol, double* P, double* TX, double* C, void* workspace, size_t workspace_bytes,
                    ct_stream_t stream) {
    if (!X || !Y || !corr || !workspace || m <= 0 || n <= 0) return CT_EINVAL;
    if (workspace_bytes < ct_prgls_workspace_bytes(m, n, 0)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* base = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    PrglsWs w;
    size_t off = prgls_layout(m, n, 0, base, &w);
    double* prior = (double*)(base + off); off += align_up((size_t)m * n * sizeof(double), 256);
    int32_t* pairs = (int32_t*)(base + off); off += align_up((size_t)n * 8 + 4, 256);
    int32_t* npairs = pairs + 2 * n;
    void* gws = base + off;
    // prior built inside with threshold 0.5 (track.py:58-70)
    int rc = ct_greedy_match(corr, m, n, 0.5f, 1, pairs, npairs, prior, gws, ct_greedy_workspace_bytes(m, n), stream);
    if (rc) return rc;
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(gauss_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, X, n, X, n, 2.0 * BETA * BETA, w.G, 0);
    LAUNCH_CHECK();
    HIPCHK(hipMemcpyAsync(w.predn, X, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    double init_sc[S_NUM] = {0.0, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};          // gamma0 = 0.1 (track.py:41)
    HIPCHK(hipMemcpyAsync(w.sc, init_sc, sizeof(init_sc), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(w.C, 0, 3 * (size_t)n * sizeof(double), st));
    hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((m + 3) / 4), dim3(256), 0, st, X, n, Y, m, (const double*)nullptr, w.rowpart,
                       (const double*)nullptr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(256), 0, st, w.rowpart, m, n, 0, w.sc, (const double*)nullptr, (const double*)nullptr);
    LAUNCH_CHECK();
    // dense M-step only: with LAMBDA ~ 1e-5 the coefficients C reach ~1e5 with massive cancellation in C.G,
    // which amplifies any truncation of the Gram matrix beyond the parity budget
    const int rank = 0;
    for (int it = 1; it < max_iteration; ++it) {
        if ((rc = em_half(w, prior, Y, m, n, X, LAMBDA, 1, vol, rank, st))) return rc;
        // T_X = X + (C G)^T recomputed from X (track.py:100)
        hipLaunchKernelGGL(apply_field_kernel, dim3((n + 3) / 4), dim3(256), 0, st, w.C, w.G, n, n, w.predn, X, 2, (double*)nullptr,
                           (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (double*)nullptr);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((m + 3) / 4), dim3(256), 0, st, w.predn, n, Y, m, w.P, w.rowpart, (const double*)nullptr);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(256), 0, st, w.rowpart, m, n, 2, w.sc, (const double*)nullptr, (const double*)nullptr);
        LAUNCH_CHECK();
    }
    if (P) HIPCHK(hipMemcpyAsync(P, w.P, (size_t)m * n * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (TX) HIPCHK(hipMemcpyAsync(TX, w.predn, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (C) HIPCHK(hipMemcpyAsync(C, w.C, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return CT_OK;
}

int ct_dist_squares(const double* ref, int n, const double* tgt, int m, double* out, ct_stream_t stream) {
    if (!ref || !tgt || !out || n <= 0 || m <= 0) return CT_EINVAL;
    const size_t tot = (size_t)m * n;
    hipLaunchKernelGGL(gauss_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref, n, tgt, m, 1.0, out, 1);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_gaussian_kernel(const double* ref, int n, const double* tgt, int m, double sigma_square, double* out, ct_stream_t stream) {
    if (!ref || !tgt || !out || n <= 0 || m <= 0) return CT_EINVAL;
    const size_t tot = (size_t)m * n;
    hipLaunchKernelGGL(gauss_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref, n, tgt, m,
                       2.0 * sigma_square, out, 0);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_estimate_posterior(const double* prior, double sigma_square, const double* pred, int n, const double* tgt, int m,
                          double ratio_outliers, double vol, double* P, ct_stream_t stream) {
    if (!prior || !pred || !tgt || !P || n <= 0 || m <= 0) return CT_EINVAL;
    hipLaunchKernelGGL(posterior_kernel, dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream, prior, pred, n, tgt, m,
                       (const double*)nullptr, 0, vol, P, sigma_square, ratio_outliers);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_solve_movements(double sigma_square, double lambda, const double* P, const double* ref, int n, const double* tgt, int m,
                       const double* G, double* C, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!P || !ref || !tgt || !G || !C || !workspace || n <= 0 || m <= 0) return CT_EINVAL;
    if (workspace_bytes < ct_prgls_workspace_bytes(m, n, 0)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    PrglsWs w;
    prgls_layout(m, n, 0, (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), &w);
    double init_sc[S_NUM] = {sigma_square, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    HIPCHK(hipMemcpyAsync(w.sc, init_sc, sizeof(init_sc), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(colstats_kernel, dim3((n + 63) / 64, CS_SEG), dim3(256), 0, st, P, tgt, m, n, w.part, (const double*)nullptr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(colstats_finish_kernel, dim3(1), dim3(256), 0, st, w.part, n, ref, lambda, w.sc, w.dvec, w.sqd, w.rhs);
    LAUNCH_CHECK();
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, G, w.sqd, w.sc, n, w.M, w.rhs, w.Spart);
    LAUNCH_CHECK();
    int rc = cholesky_solve(w, n, st);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(C, w.C, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return CT_OK;
}

int ct_gram_apply(double* pred, int l, const double* inter, int n, const double* C, double beta, ct_stream_t stream) {
    if (!pred || !inter || !C || l <= 0 || n <= 0) return CT_EINVAL;
    hipLaunchKernelGGL(gram_apply_kernel, dim3((l + 3) / 4), dim3(256), 0, (hipStream_t)stream, pred, l, inter, n, C, 2.0 * beta * beta);
    LAUNCH_CHECK();
    return CT_OK;
}

// ------------------------------------------------------------------------------------------------ legacy prediction chain
// Tracker._predict_pos_once (tracker.py:1193-1222) = _fit_ffn_prgls (:1224-1254: `reps` x [initial_matching_quick -> pr_gls_quick with
// BETA * 0.8^i, each repetition starting from the previous one's transformed points]) followed by _predict_one_rep (:1269-1289) for
// every repetition.  One call per source volume: the ~500 launches of the chain are enqueued from C, so the host threads that drive the
// <= 20 independent source volumes of an ensemble prediction (tracker.py:1499-1506) are not serialised by the Python interpreter lock
// (composed from Python the ensemble saturated at ~3 ms per source volume whatever the number of chains).
static size_t legacy_predict_layout(int n, int m, int reps, int k, size_t off[8]) {
    size_t o = 0;
    const size_t fw = (size_t)(3 * k + 1);
    off[0] = o; o += align_up((size_t)n * fw * sizeof(float), 256);                 // feat_ref
    off[1] = o; o += align_up((size_t)m * fw * sizeof(float), 256);                 // feat_tgt
    off[2] = o; o += align_up((size_t)m * n * sizeof(float), 256);                  // corr
    off[3] = o; o += align_up((size_t)(reps + 1) * n * 3 * sizeof(double), 256);    // inter[0..reps]
    off[4] = o; o += align_up((size_t)reps * 3 * n * sizeof(double), 256);          // C[0..reps-1]
    off[5] = o; o += align_up(ct_ffn_workspace_bytes(n, m), 256);                   // FFN scratch
    off[6] = o; o += align_up(ct_prgls_workspace_bytes(m, n, 0), 256);              // PR-GLS scratch
    return o;
}

size_t ct_legacy_predict_workspace_bytes(int n, int m, int reps, int k_ptrs) {
    if (n <= 0 || m <= 0 || reps <= 0 || k_ptrs <= 0) return 0;
    size_t off[8];
    return legacy_predict_layout(n, m, reps, k_ptrs, off) + 512;
}

int ct_legacy_predict_pos(ct_ffn_t* ffn, const double* seg_pre, int n, const double* seg_tgt, int m, const double* tracked_pre, int l,
                          double beta, double lambda, int max_iteration, int reps, int k_ptrs, double* pred_out, double* C_out,
                          double* inter_out, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!ffn || !seg_pre || !seg_tgt || !workspace || n <= 0 || m <= 0 || reps <= 0 || k_ptrs <= 0 || l < 0 || (l > 0 && (!tracked_pre || !pred_out)))
        return CT_EINVAL;
    if (workspace_bytes < ct_legacy_predict_workspace_bytes(n, m, reps, k_ptrs)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* base = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    size_t off[8];
    legacy_predict_layout(n, m, reps, k_ptrs, off);
    float* feat_ref = (float*)(base + off[0]); float* feat_tgt = (float*)(base + off[1]); float* corr = (float*)(base + off[2]);
    double* inter = (double*)(base + off[3]); double* Cs = (double*)(base + off[4]);
    void* fws = base + off[5]; void* pws = base + off[6];
    const size_t n3 = (size_t)n * 3;
    int rc;
    HIPCHK(hipMemcpyAsync(inter, seg_pre, n3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if ((rc = ct_knn_features(seg_tgt, m, k_ptrs, feat_tgt, stream))) return rc;      // the target's features are the same in every repetition
    for (int i = 0; i < reps; ++i) {
        const double b = beta * pow(0.8, (double)i);
        if ((rc = ct_knn_features(inter + i * n3, n, k_ptrs, feat_ref, stream))) return rc;
        if ((rc = ct_ffn_pairgrid(ffn, feat_ref, n, feat_tgt, m, corr, fws, ct_ffn_workspace_bytes(n, m), stream))) return rc;
        if ((rc = ct_prgls_legacy(inter + i * n3, n, seg_tgt, m, corr, b, max_iteration, lambda, 1e8, nullptr, inter + (i + 1) * n3,
                                  Cs + (size_t)i * n3, pws, ct_prgls_workspace_bytes(m, n, 0), stream)))
            return rc;
    }
    if (l > 0) {
        HIPCHK(hipMemcpyAsync(pred_out, tracked_pre, (size_t)l * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
        for (int i = 0; i < reps; ++i)
            if ((rc = ct_gram_apply(pred_out, l, inter + i * n3, n, Cs + (size_t)i * n3, beta * pow(0.8, (double)i), stream))) return rc;
    }
    if (C_out) HIPCHK(hipMemcpyAsync(C_out, Cs, (size_t)reps * n3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (inter_out) HIPCHK(hipMemcpyAsync(inter_out, inter, (size_t)reps * n3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    return CT_OK;
}

// FFN scores + greedy prior (the front half of TrackerLite.predict_cell_positions, trackerlite.py:83-91: initial_matching_ffn ->
// simple_match) for B independent problems as one chain of launches; ragged reference AND target sets.  prior_out[b] [dev] fp64
// [m[b]][n[b]].  Bit-identical to ct_knn_features x 2 + ct_ffn_pairgrid + ct_greedy_match per problem.  Synchronises `stream`.
struct MatchFrontSlab { size_t ref, tgt, featr, featt, Hr, Ht, U, V, corr, pairs, gd, gd_keys, gd_ctr, gd_words, prior, size; };
static MatchFrontSlab match_front_slab(int nmax, int mmax, int k) {
    MatchFrontSlab L{}; size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
    const size_t n = (size_t)nmax, m = (size_t)mmax, fw = (size_t)(3 * k + 1);
    L.ref = take(n * 24); L.tgt = take(m * 24); L.featr = take(n * fw * 4); L.featt = take(m * fw * 4);
    L.Hr = take(n * HID * 4); L.Ht = take(m * HID * 4); L.U = take(n * HID * 4); L.V = take(m * HID * 4); L.corr = take(m * n * 4);
    L.pairs = take(n * 8 + 8);
    L.gd = o;
    take(m); take(n); take(m * 4); take(m * 4); take(m * 4); take(n * 4);
    L.gd_keys = take((m > n ? m : n) * 8); L.gd_ctr = take(256);
    L.gd_words = (o - L.gd) / 4;
    L.prior = take(m * n * 8);
    L.size = o;
    return L;
}
size_t ct_match_front_batched_workspace_bytes(int B, int nmax, int mmax, int k_ptrs) {
    if (B <= 0 || nmax <= 0 || mmax <= 0 || k_ptrs <= 0) return 0;
    return (size_t)B * match_front_slab(nmax, mmax, k_ptrs).size + 3 * align_up((size_t)B * 16, 256) + 512;
}
int ct_match_front_batched(ct_ffn_t* ffn, int B, const double* const* ref, const int* n, const double* const* tgt, const int* m, int k_ptrs,
                           float threshold, int mode, double* const* prior_out, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!ffn || B <= 0 || !ref || !n || !tgt || !m || !prior_out || !workspace || k_ptrs <= 0 || (mode != 0 && mode != 1)) return CT_EINVAL;
    int nmax = 0, mmax = 0;
    for (int b = 0; b < B; ++b) {
        if (!ref[b] || !tgt[b] || !prior_out[b] || n[b] <= k_ptrs || m[b] <= k_ptrs) return CT_EINVAL;
        nmax = n[b] > nmax ? n[b] : nmax; mmax = m[b] > mmax ? m[b] : mmax;
    }
    if (nmax > KNN_MAXN || mmax > KNN_MAXN || (nmax < mmax ? nmax : mmax) > 16384) return CT_ESHAPE;
    if (workspace_bytes < ct_match_front_batched_workspace_bytes(B, nmax, mmax, k_ptrs)) return CT_EWORKSPACE;
    DeviceGuard dg(ffn->device);
    if (dg.err != hipSuccess) return (int)dg.err;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* base = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const MatchFrontSlab L = match_front_slab(nmax, mmax, k_ptrs);
    unsigned char* shared = base + (size_t)B * L.size;
    int* d_dims = (int*)shared; shared += align_up((size_t)B * 16, 256);      // {m, n, 0, 1}: problem sizes; [3] = 1: per-problem target half
    int* d_dimsT = (int*)shared; shared += align_up((size_t)B * 16, 256);     // {n, m, ..}: the same table seen from the target set
    int* d_ctr = (int*)shared;
    std::vector<int> hd((size_t)B * 8, 0), hctr((size_t)B * 4, 0);
    for (int b = 0; b < B; ++b) { hd[4 * b] = m[b]; hd[4 * b + 1] = n[b]; hd[4 * b + 3] = 1; hd[4 * B + 4 * b] = n[b]; hd[4 * B + 4 * b + 1] = m[b]; }
    HIPCHK(hipMemcpyAsync(d_dims, hd.data(), (size_t)B * 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_dimsT, hd.data() + 4 * B, (size_t)B * 16, hipMemcpyHostToDevice, st));
    const Bt bt{L.size / sizeof(double), d_dims}, btT{L.size / sizeof(double), d_dimsT};
    const unsigned zB = (unsigned)B;
    auto at = [&](size_t off) { return base + off; };
    for (int b = 0; b < B; ++b) {
        unsigned char* sb = base + (size_t)b * L.size;
        HIPCHK(hipMemcpyAsync(sb + L.ref, ref[b], (size_t)n[b] * 24, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(sb + L.tgt, tgt[b], (size_t)m[b] * 24, hipMemcpyDeviceToDevice, st));
    }
    ENSURE_BIG_LDS(gd_finalize_kernel);
    const float* W = ffn->d_w;
    float* featr = (float*)at(L.featr); float* featt = (float*)at(L.featt); float* Hr = (float*)at(L.Hr); float* Ht = (float*)at(L.Ht);
    float* U = (float*)at(L.U); float* V = (float*)at(L.V); float* corr = (float*)at(L.corr);
    hipLaunchKernelGGL(knn_features_kernel, dim3(nmax, 1, zB), dim3(64), 0, st, (const double*)at(L.ref), nmax, k_ptrs, featr, bt);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_features_kernel, dim3(mmax, 1, zB), dim3(64), 0, st, (const double*)at(L.tgt), mmax, k_ptrs, featt, btT);
    LAUNCH_CHECK();
    { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, featr, FEAT, W + ffn->o_w1, Hr, nmax, HID, FEAT, W + ffn->o_bn1, bt); else hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, featr, FEAT, W + ffn->o_w1, Hr, nmax, HID, FEAT, W + ffn->o_bn1, bt); }
    LAUNCH_CHECK();
    { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((HID + 63) / 64, (mmax + 63) / 64, zB), dim3(256), 0, st, featt, FEAT, W + ffn->o_w1, Ht, mmax, HID, FEAT, W + ffn->o_bn1, btT); else hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((HID + 63) / 64, (mmax + 63) / 64, zB), dim3(256), 0, st, featt, FEAT, W + ffn->o_w1, Ht, mmax, HID, FEAT, W + ffn->o_bn1, btT); }
    LAUNCH_CHECK();
    { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, Hr, HID, W + ffn->o_w2, U, nmax, HID, HID, (const float*)nullptr, bt); else hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, Hr, HID, W + ffn->o_w2, U, nmax, HID, HID, (const float*)nullptr, bt); }
    LAUNCH_CHECK();
    { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((HID + 63) / 64, (mmax + 63) / 64, zB), dim3(256), 0, st, Ht, HID, W + ffn->o_w2 + (size_t)HID * HID, V, mmax, HID, HID,
                       (const float*)nullptr, btT); else hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((HID + 63) / 64, (mmax + 63) / 64, zB), dim3(256), 0, st, Ht, HID, W + ffn->o_w2 + (size_t)HID * HID, V, mmax, HID, HID,
                       (const float*)nullptr, btT); }
    LAUNCH_CHECK();
    hipLaunchKernelGGL(ffn_pair_kernel, dim3((nmax + 31) / 32, (mmax + 31) / 32, zB), dim3(256), 0, st, U, nmax, V, mmax, W + ffn->o_bn2, W + ffn->o_w3, ffn->b3, corr, bt);
    LAUNCH_CHECK();
    unsigned char* gd = at(L.gd);
    unsigned char* row_used = gd; size_t go = align_up((size_t)mmax, 256);
    unsigned char* col_used = gd + go; go += align_up((size_t)nmax, 256);
    float* rowval = (float*)(gd + go); go += align_up((size_t)mmax * 4, 256);
    int* rowcol = (int*)(gd + go); go += align_up((size_t)mmax * 4, 256);
    int* row_match = (int*)(gd + go); go += align_up((size_t)mmax * 4, 256);
    int* colrow = (int*)(gd + go);
    unsigned long long* keys = (unsigned long long*)at(L.gd_keys); int* ctr = (int*)at(L.gd_ctr);
    int32_t* pairs = (int32_t*)at(L.pairs); int32_t* npairs = pairs + 2 * nmax;
    double* prior = (double*)at(L.prior);
    hipLaunchKernelGGL(bt_zero_kernel, dim3(8, 1, zB), dim3(256), 0, st, gd, (int)L.gd_words, bt);
    LAUNCH_CHECK();
    const int nrb = (mmax + GD_ROWLANES_BATCH - 1) / GD_ROWLANES_BATCH, ncb = (nmax + 63) / 64;
    const int max_rounds = (mmax < nmax ? mmax : nmax) + 1;
    int maxcnt = 0;
    for (int done_rounds = 0; done_rounds < max_rounds;) {
        const int chunk = done_rounds == 0 ? 12 : 8;
        for (int k = 0; k < chunk; ++k) {
            const int round = done_rounds + k;
            hipLaunchKernelGGL(gd_best_kernel<GD_ROWLANES_BATCH>, dim3(nrb + ncb, 1, zB), dim3(64 * GD_ROWLANES_BATCH), 0, st, corr, mmax, nmax, row_used, col_used, rowval, rowcol, colrow,
                               ctr, round, nrb, bt);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(gd_accept_kernel, dim3((mmax + 255) / 256, 1, zB), dim3(256), 0, st, mmax, nmax, threshold, rowval, rowcol, colrow, row_used,
                               col_used, keys, ctr, round, bt);
            LAUNCH_CHECK();
        }
        done_rounds += chunk;
        hipLaunchKernelGGL(bt_gather_ctr_kernel, dim3((B + 63) / 64), dim3(64), 0, st, ctr, bt.stride, B, d_ctr);
        LAUNCH_CHECK();
        HIPCHK(hipMemcpyAsync(hctr.data(), d_ctr, hctr.size() * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        bool all_done = true; maxcnt = 0;
        for (int b = 0; b < B; ++b) {
            if (!(hctr[4 * b + GD_DONE] || hctr[4 * b + GD_NEW + ((done_rounds - 1) & 1)] == 0)) all_done = false;
            maxcnt = hctr[4 * b + GD_COUNT] > maxcnt ? hctr[4 * b + GD_COUNT] : maxcnt;
        }
        if (all_done) break;
    }
    {
        int p2 = 1; while (p2 < maxcnt) p2 <<= 1;
        hipLaunchKernelGGL(gd_finalize_kernel, dim3(1, 1, zB), dim3(1024), (size_t)p2 * 8, st, keys, ctr, nmax, pairs, npairs, bt);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(row_match_kernel, dim3((mmax + 255) / 256, 1, zB), dim3(256), 0, st, row_match, mmax, bt);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(row_match_set_kernel, dim3((nmax + 255) / 256, 1, zB), dim3(256), 0, st, row_match, pairs, npairs, bt);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(prior_fill_kernel, dim3((unsigned)(((size_t)mmax * nmax + 255) / 256), 1, zB), dim3(256), 0, st, prior, mmax, nmax, mode, pairs, npairs,
                       row_match, bt);
    LAUNCH_CHECK();
    for (int b = 0; b < B; ++b)
        HIPCHK(hipMemcpyAsync(prior_out[b], base + (size_t)b * L.size + L.prior, (size_t)m[b] * n[b] * sizeof(double), hipMemcpyDeviceToDevice, st));
    return CT_OK;
}

// The same chain for B source volumes at once: problem b = blockIdx.z of every kernel, its buffers one slab further (ragged n[b]; the
// target set, the tracked-point count and all parameters are shared).  Requires max n <= 132 (the single-workgroup dense M-step; the
// caller falls back to ct_legacy_predict_pos per volume otherwise).  ~500 launches per ensemble step instead of ~500 per source volume:
// the chain is bound by the dispatch rate of dependent tiny kernels, not by their work.  Bit-identical to B separate calls.
struct LegacySlab {
    size_t inter, Cs, feat, Hr, U, corr, prior, pairs, gd, G, P, part, dvec, sqd, rhs, C, predn, rowpart, sc, tgt, pred, size;
    size_t gd_keys, gd_ctr, gd_words;
};
static LegacySlab legacy_slab(int nmax, int m, int l, int reps, int k) {
    LegacySlab L{}; size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
    const size_t n = (size_t)nmax, fw = (size_t)(3 * k + 1);
    L.inter = take((size_t)(reps + 1) * n * 3 * 8); L.Cs = take((size_t)reps * 3 * n * 8);
    L.feat = take(n * fw * 4); L.Hr = take(n * HID * 4); L.U = take(n * HID * 4); L.corr = take((size_t)m * n * 4);
    L.prior = take((size_t)m * n * 8); L.pairs = take(n * 8 + 8);
    L.gd = o;                                                  // greedy scratch, zeroed per repetition: flags, tables, keys, counters
    take((size_t)m); take(n); take((size_t)m * 4); take((size_t)m * 4); take((size_t)m * 4); take(n * 4);
    L.gd_keys = take((size_t)(m > nmax ? m : nmax) * 8); L.gd_ctr = take(256);
    L.gd_words = (o - L.gd) / 4;
    L.G = take(n * n * 8); L.P = take((size_t)m * n * 8); L.part = take((size_t)CS_SEG * 4 * n * 8); L.dvec = take(n * 8); L.sqd = take(n * 8);
    L.rhs = take(3 * n * 8); L.C = take(3 * n * 8); L.predn = take(3 * n * 8); L.rowpart = take((size_t)m * 8); L.sc = take(S_NUM * 8);
    L.tgt = take((size_t)m * 3 * 8); L.pred = take((size_t)(l > 0 ? l : 1) * 3 * 8);
    L.size = o;
    return L;
}
static size_t legacy_shared_bytes(int B, int m, int k) {
    return align_up((size_t)m * (3 * k + 1) * 4, 256) + 2 * align_up((size_t)m * HID * 4, 256) + align_up((size_t)B * 16, 256) + align_up((size_t)B * 16, 256);
}

size_t ct_legacy_predict_batched_workspace_bytes(int B, int nmax, int m, int l, int reps, int k_ptrs) {
    if (B <= 0 || nmax <= 0 || m <= 0 || reps <= 0 || k_ptrs <= 0 || l < 0) return 0;
    return (size_t)B * legacy_slab(nmax, m, l, reps, k_ptrs).size + legacy_shared_bytes(B, m, k_ptrs) + 512;
}

int ct_legacy_predict_pos_batched(ct_ffn_t* ffn, int B, const double* const* seg_pre, const int* n, const double* seg_tgt, int m,
                                  const double* const* tracked_pre, int l, double beta, double lambda, int max_iteration, int reps,
                                  int k_ptrs, double* pred_out, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!ffn || B <= 0 || !seg_pre || !n || !seg_tgt || m <= 0 || !workspace || reps <= 0 || k_ptrs <= 0 || l <= 0 || !tracked_pre || !pred_out)
        return CT_EINVAL;
    int nmax = 0;
    for (int b = 0; b < B; ++b) { if (n[b] <= k_ptrs || !seg_pre[b] || !tracked_pre[b]) return CT_EINVAL; nmax = n[b] > nmax ? n[b] : nmax; }
    if (nmax > DS_MAXN || m <= k_ptrs) return CT_ESHAPE;
    if (workspace_bytes < ct_legacy_predict_batched_workspace_bytes(B, nmax, m, l, reps, k_ptrs)) return CT_EWORKSPACE;
    DeviceGuard dg(ffn->device);
    if (dg.err != hipSuccess) return (int)dg.err;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* base = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const LegacySlab L = legacy_slab(nmax, m, l, reps, k_ptrs);
    unsigned char* shared = base + (size_t)B * L.size;
    float* feat_tgt = (float*)shared; shared += align_up((size_t)m * (3 * k_ptrs + 1) * 4, 256);
    float* Ht = (float*)shared; shared += align_up((size_t)m * HID * 4, 256);
    float* V = (float*)shared; shared += align_up((size_t)m * HID * 4, 256);
    int* d_dims = (int*)shared; shared += align_up((size_t)B * 16, 256);
    int* d_ctr = (int*)shared;
    std::vector<int> hdims((size_t)B * 4, 0), hctr((size_t)B * 4, 0);
    for (int b = 0; b < B; ++b) { hdims[4 * b] = m; hdims[4 * b + 1] = n[b]; hdims[4 * b + 2] = l; }
    HIPCHK(hipMemcpyAsync(d_dims, hdims.data(), hdims.size() * sizeof(int), hipMemcpyHostToDevice, st));
    const Bt bt{L.size / sizeof(double), d_dims};
    const unsigned zB = (unsigned)B;
    auto at = [&](size_t off) { return base + off; };        // problem 0's copy of a slab member
    const size_t n3max = (size_t)nmax * 3;
    for (int b = 0; b < B; ++b) {
        unsigned char* sb = base + (size_t)b * L.size;
        HIPCHK(hipMemcpyAsync(sb + L.inter, seg_pre[b], (size_t)n[b] * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(sb + L.pred, tracked_pre[b], (size_t)l * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(sb + L.tgt, seg_tgt, (size_t)m * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    ENSURE_BIG_LDS(gd_finalize_kernel);
    ENSURE_BIG_LDS(dense_small_solve_kernel);
    const float* W = ffn->d_w;
    // target half of the FFN: once for the whole batch
    hipLaunchKernelGGL(knn_features_kernel, dim3(m), dim3(64), 0, st, seg_tgt, m, k_ptrs, feat_tgt, Bt{0, nullptr});
    LAUNCH_CHECK();
    int rc;
    if ((rc = gemm(feat_tgt, FEAT, W + ffn->o_w1, Ht, m, HID, FEAT, W + ffn->o_bn1, st))) return rc;
    if ((rc = gemm(Ht, HID, W + ffn->o_w2 + (size_t)HID * HID, V, m, HID, HID, nullptr, st))) return rc;
    const int nrb = (m + GD_ROWLANES_BATCH - 1) / GD_ROWLANES_BATCH, ncb = (nmax + 63) / 64;
    for (int i = 0; i < reps; ++i) {
        const double bi = beta * pow(0.8, (double)i);
        double* X = (double*)at(L.inter) + (size_t)i * n3max;               // this repetition's reference points (slab-relative)
        float* feat = (float*)at(L.feat); float* Hr = (float*)at(L.Hr); float* U = (float*)at(L.U); float* corr = (float*)at(L.corr);
        // initial_matching_quick (track.py:117-178): features -> FFN on all pairs
        hipLaunchKernelGGL(knn_features_kernel, dim3(nmax, 1, zB), dim3(64), 0, st, X, nmax, k_ptrs, feat, bt);
        LAUNCH_CHECK();
        { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, feat, FEAT, W + ffn->o_w1, Hr, nmax, HID, FEAT,
                           W + ffn->o_bn1, bt); else hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, feat, FEAT, W + ffn->o_w1, Hr, nmax, HID, FEAT,
                           W + ffn->o_bn1, bt); }
        LAUNCH_CHECK();
        { if (gemm_valu()) hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, Hr, HID, W + ffn->o_w2, U, nmax, HID, HID,
                           (const float*)nullptr, bt); else hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((HID + 63) / 64, (nmax + 63) / 64, zB), dim3(256), 0, st, Hr, HID, W + ffn->o_w2, U, nmax, HID, HID,
                           (const float*)nullptr, bt); }
        LAUNCH_CHECK();
        hipLaunchKernelGGL(ffn_pair_kernel, dim3((nmax + 31) / 32, (m + 31) / 32, zB), dim3(256), 0, st, U, nmax, V, m, W + ffn->o_bn2, W + ffn->o_w3,
                           ffn->b3, corr, bt);
        LAUNCH_CHECK();
        // pr_gls_quick (track.py:11-114): greedy prior with threshold 0.5 ...
        unsigned char* gd = at(L.gd);
        unsigned char* row_used = gd; size_t go = align_up((size_t)m, 256);
        unsigned char* col_used = gd + go; go += align_up((size_t)nmax, 256);
        float* rowval = (float*)(gd + go); go += align_up((size_t)m * 4, 256);
        int* rowcol = (int*)(gd + go); go += align_up((size_t)m * 4, 256);
        int* row_match = (int*)(gd + go); go += align_up((size_t)m * 4, 256);
        int* colrow = (int*)(gd + go);
        unsigned long long* keys = (unsigned long long*)at(L.gd_keys); int* ctr = (int*)at(L.gd_ctr);
        int32_t* pairs = (int32_t*)at(L.pairs); int32_t* npairs = pairs + 2 * nmax;
        double* prior = (double*)at(L.prior);
        hipLaunchKernelGGL(bt_zero_kernel, dim3(8, 1, zB), dim3(256), 0, st, gd, (int)L.gd_words, bt);
        LAUNCH_CHECK();
        const int max_rounds = (m < nmax ? m : nmax) + 1;
        int maxcnt = 0;
        for (int done_rounds = 0; done_rounds < max_rounds;) {
            const int chunk = done_rounds == 0 ? 12 : 8;
            for (int k = 0; k < chunk; ++k) {
                const int round = done_rounds + k;
                hipLaunchKernelGGL(gd_best_kernel<GD_ROWLANES_BATCH>, dim3(nrb + ncb, 1, zB), dim3(64 * GD_ROWLANES_BATCH), 0, st, corr, m, nmax, row_used, col_used, rowval, rowcol,
                                   colrow, ctr, round, nrb, bt);
                LAUNCH_CHECK();
                hipLaunchKernelGGL(gd_accept_kernel, dim3((m + 255) / 256, 1, zB), dim3(256), 0, st, m, nmax, 0.5f, rowval, rowcol, colrow, row_used,
                                   col_used, keys, ctr, round, bt);
                LAUNCH_CHECK();
            }
            done_rounds += chunk;
            hipLaunchKernelGGL(bt_gather_ctr_kernel, dim3((B + 63) / 64), dim3(64), 0, st, ctr, bt.stride, B, d_ctr);
            LAUNCH_CHECK();
            HIPCHK(hipMemcpyAsync(hctr.data(), d_ctr, hctr.size() * sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            bool all_done = true; maxcnt = 0;
            for (int b = 0; b < B; ++b) {
                if (!(hctr[4 * b + GD_DONE] || hctr[4 * b + GD_NEW + ((done_rounds - 1) & 1)] == 0)) all_done = false;
                maxcnt = hctr[4 * b + GD_COUNT] > maxcnt ? hctr[4 * b + GD_COUNT] : maxcnt;
            }
            if (all_done) break;
        }
        {
            int p2 = 1; while (p2 < maxcnt) p2 <<= 1;
            hipLaunchKernelGGL(gd_finalize_kernel, dim3(1, 1, zB), dim3(1024), (size_t)p2 * 8, st, keys, ctr, nmax, pairs, npairs, bt);
            LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(row_match_kernel, dim3((m + 255) / 256, 1, zB), dim3(256), 0, st, row_match, m, bt);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(row_match_set_kernel, dim3((nmax + 255) / 256, 1, zB), dim3(256), 0, st, row_match, pairs, npairs, bt);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(prior_fill_kernel, dim3((unsigned)(((size_t)m * nmax + 255) / 256), 1, zB), dim3(256), 0, st, prior, m, nmax, 1, pairs, npairs,
                           row_match, bt);
        LAUNCH_CHECK();
        // ... then max_iteration - 1 EM iterations with the dense M-step
        double* G = (double*)at(L.G); double* P = (double*)at(L.P); double* part = (double*)at(L.part); double* dvec = (double*)at(L.dvec);
        double* sqd = (double*)at(L.sqd); double* rhs = (double*)at(L.rhs); double* Cc = (double*)at(L.C); double* predn = (double*)at(L.predn);
        double* rowpart = (double*)at(L.rowpart); double* sc = (double*)at(L.sc); double* Y = (double*)at(L.tgt);
        hipLaunchKernelGGL(gauss_kernel, dim3((unsigned)(((size_t)nmax * nmax + 255) / 256), 1, zB), dim3(256), 0, st, X, nmax, X, nmax, 2.0 * bi * bi, G, 0, bt);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(bt_copy3_kernel, dim3(2, 1, zB), dim3(256), 0, st, predn, X, 1, bt);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(bt_legacy_init_kernel, dim3(1, 1, zB), dim3(256), 0, st, sc, Cc, bt);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((m + 3) / 4, 1, zB), dim3(256), 0, st, X, nmax, Y, m, (const double*)nullptr, rowpart,
                           (const double*)nullptr, bt);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(scalars_kernel, dim3(1, 1, zB), dim3(256), 0, st, rowpart, m, nmax, 0, sc, (const double*)nullptr, (const double*)nullptr,
                           (int*)nullptr, bt);
        LAUNCH_CHECK();
        const size_t lds = (size_t)(nmax + 3) * (nmax | 1) * sizeof(double);
        for (int it = 1; it < max_iteration; ++it) {
            hipLaunchKernelGGL(posterior_kernel, dim3((m + 3) / 4, 1, zB), dim3(256), 0, st, prior, predn, nmax, Y, m, sc, 1, 1e8, P, 0.0, 0.0, bt);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(colstats_kernel, dim3((nmax + 63) / 64, CS_SEG, zB), dim3(256), 0, st, P, Y, m, nmax, part, sc, bt);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(dense_small_solve_kernel, dim3(1, 1, zB), dim3(256), lds, st, part, nmax, X, G, lambda, sc, dvec, sqd, rhs, Cc, bt);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(apply_field_kernel, dim3((nmax + 3) / 4, 1, zB), dim3(256), 0, st, Cc, G, nmax, nmax, predn, X, 2, (double*)nullptr,
                               (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, bt);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(dist2_rowsum_kernel, dim3((m + 3) / 4, 1, zB), dim3(256), 0, st, predn, nmax, Y, m, P, rowpart, (const double*)nullptr, bt);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(scalars_kernel, dim3(1, 1, zB), dim3(256), 0, st, rowpart, m, nmax, 2, sc, (const double*)nullptr, (const double*)nullptr,
                               (int*)nullptr, bt);
            LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(bt_copy3_kernel, dim3(2, 1, zB), dim3(256), 0, st, X + n3max, predn, 1, bt);                    // next repetition starts from T_X
        LAUNCH_CHECK();
        hipLaunchKernelGGL(bt_copy3_kernel, dim3(2, 1, zB), dim3(256), 0, st, (double*)at(L.Cs) + (size_t)i * n3max, Cc, 1, bt);
        LAUNCH_CHECK();
    }
    // _predict_one_rep (tracker.py:1269-1289) for every repetition
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(gram_apply_kernel, dim3((l + 3) / 4, 1, zB), dim3(256), 0, st, (double*)at(L.pred), l, (double*)at(L.inter) + (size_t)i * n3max, nmax,
                           (double*)at(L.Cs) + (size_t)i * n3max, 2.0 * (beta * pow(0.8, (double)i)) * (beta * pow(0.8, (double)i)), bt);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bt_pack_kernel, dim3(2, 1, zB), dim3(256), 0, st, (const double*)at(L.pred), bt.stride, 3 * l, pred_out);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_trim_mean(const double* stack, int k, int n3, double cut, double* out, ct_stream_t stream) {
    if (!stack || !out || k <= 0 || n3 <= 0 || cut < 0.0 || cut >= 0.5) return CT_EINVAL;
    if (k > TM_MAXK) return CT_ESHAPE;
    const int lo = (int)(cut * k);
    hipLaunchKernelGGL(trim_mean_kernel, dim3((n3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, stack, k, n3, lo, out);
    LAUNCH_CHECK();
    return CT_OK;
}

}  // extern "C"
