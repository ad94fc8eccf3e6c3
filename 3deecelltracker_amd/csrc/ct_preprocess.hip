// ct_preprocess.hip -- local contrast normalisation that precedes the U-Net every frame.
//
// What it replaces (reference CellTracker/preprocess.py):
//   :170-188 _normalize_image   image - np.median(image); clamp at 0; lcn_gpu(..., filter_size=(27, 27, 1))
//   :136-167 lcn_gpu            avg = box(x)/V; std = sqrt(box((x-avg)^2)/V); (x-avg)/(std+noise)   [Keras Conv3D, zero pad]
//   :85-114  lcn_cpu            the same with scipy 'reflect' borders
//
// HBM-bound streaming work: an exact median by 8-bit radix select (2 passes for uint16, 4 for float32; LDS
// histograms, no sort), then separable 1-D box sums (27 + 27 terms instead of 729; double accumulation, fp32 storage)
// and two element-wise kernels.  Everything stays on the device; the median never visits the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/ctamd.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ uint32_t key_of(const void* data, int dtype, size_t i) {
    if (dtype == 0) return (uint32_t)reinterpret_cast<const uint16_t*>(data)[i];
    uint32_t u = reinterpret_cast<const uint32_t*>(data)[i];          // float32 -> order-preserving key
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ double value_of_key(uint32_t k, int dtype) {
    if (dtype == 0) return (double)k;
    const uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return (double)__uint_as_float(u);
}

// select state (device): [0] prefix value, [1] prefix mask, [2..3] remaining rank (64-bit)
struct SelState { uint32_t prefix, mask; unsigned long long rank; };

// Both order statistics of np.median ((n-1)/2 and n/2) are selected in the same passes: st[0] / st[1] share one histogram
// while their prefixes agree (almost always -- they are neighbours in sorted order) and get separate ones once they differ.
// hist: [2][256].  A wave whose 64 keys fall into one bin (background voxels in the high-byte pass) adds once.
__global__ __launch_bounds__(256) void radix_hist_kernel(const void* __restrict__ data, int dtype, size_t n, int shift,
                                                         const SelState* __restrict__ st, unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[2][256];
    h[0][threadIdx.x] = 0; h[1][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t p0 = st[0].prefix, p1 = st[1].prefix, mask = st[0].mask;      // the masks are always equal
    const bool same = p0 == p1;
    const size_t nround = (n + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);
    for (size_t rnd = 0; rnd < nround; ++rnd) {
        const size_t i = (rnd * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        const bool live = i < n;
        const uint32_t k = live ? key_of(data, dtype, i) : 0u;
        const int d = (int)((k >> shift) & 0xFF);
        const bool m0 = live && (k & mask) == p0, m1 = live && !same && (k & mask) == p1;
        const unsigned long long b0 = __ballot(m0);
        if (b0) {
            const int lead = __ffsll((long long)b0) - 1;
            const int dl = __shfl(d, lead);
            if (__ballot(m0 && d == dl) == b0) { if ((int)(threadIdx.x & 63) == lead) atomicAdd(&h[0][dl], (unsigned)__popcll(b0)); }
            else if (m0) atomicAdd(&h[0][d], 1u);
        }
        if (m1) atomicAdd(&h[1][d], 1u);
    }
    __syncthreads();
    if (h[0][threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[0][threadIdx.x]);
    if (h[1][threadIdx.x]) atomicAdd(&hist[256 + threadIdx.x], h[1][threadIdx.x]);
}

__global__ void radix_pick_kernel(unsigned int* __restrict__ hist, int shift, SelState* __restrict__ st) {
    if (threadIdx.x == 0) {
        const bool same = st[0].prefix == st[1].prefix;
        for (int q = 0; q < 2; ++q) {
            const unsigned int* hq = hist + ((q == 1 && !same) ? 256 : 0);
            unsigned long long r = st[q].rank, acc = 0; int d = 0;
            for (; d < 256; ++d) { if (acc + hq[d] > r) break; acc += hq[d]; }
            if (d > 255) d = 255;
            st[q].rank = r - acc;
            st[q].prefix |= ((uint32_t)d << shift);
            st[q].mask |= (0xFFu << shift);
        }
    }
    __syncthreads();
    hist[threadIdx.x] = 0; hist[256 + threadIdx.x] = 0;          // ready for the next pass (blockDim = 256)
}

__global__ void median_finish_kernel(const SelState* __restrict__ lo, const SelState* __restrict__ hi, int dtype,
                                     double* __restrict__ median) {
    *median = 0.5 * (value_of_key(lo->prefix, dtype) + value_of_key(hi->prefix, dtype));      // np.median: mean of the two middles
}

// x = max(img - median, 0)   (preprocess.py:186-187)   or plain conversion when median == nullptr
__global__ __launch_bounds__(256) void prep_kernel(const void* __restrict__ img, int dtype, size_t n, const double* __restrict__ median,
                                                   float* __restrict__ x) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = dtype == 0 ? (double)reinterpret_cast<const uint16_t*>(img)[i] : (double)reinterpret_cast<const float*>(img)[i];
    if (median) { v -= *median; if (v < 0.0) v = 0.0; }
    x[i] = (float)v;
}

// 1-D centred box sum along `axis` of a [X][Y][Z] array (z fastest); mode 0 = zero padding, 1 = scipy 'reflect'
__global__ __launch_bounds__(256) void box1d_kernel(const float* __restrict__ in, float* __restrict__ out, int X, int Y, int Z,
                                                    int axis, int half, int mode) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)X * Y * Z;
    if (i >= n) return;
    const int z = (int)(i % Z); const int y = (int)((i / Z) % Y); const int x = (int)(i / ((size_t)Y * Z));
    const int len = axis == 0 ? X : (axis == 1 ? Y : Z);
    const int pos = axis == 0 ? x : (axis == 1 ? y : z);
    const size_t stride = axis == 0 ? (size_t)Y * Z : (axis == 1 ? (size_t)Z : 1);
    const size_t base = i - (size_t)pos * stride;
    double acc = 0.0;
    for (int d = -half; d <= half; ++d) {
        int q = pos + d;
        if (mode == 0) { if (q < 0 || q >= len) continue; }
        else {                                       // d c b a | a b c d | d c b a
            const int period = 2 * len;
            q %= period; if (q < 0) q += period;
            if (q >= len) q = period - 1 - q;
        }
        acc += (double)in[base + (size_t)q * stride];
    }
    out[i] = (float)acc;
}

// Sliding-window form of box1d_kernel for the x and y axes (stride >= Z): one thread walks a segment of one line, so an
// output costs ~2.4 reads instead of `2 half + 1` (27 by default).  The window sum is carried in double with an
// error-free TwoSum compensation term, i.e. it equals the directly accumulated double sum to ~2^-100 of the largest
// partial sum; consecutive threads own consecutive lines (z fastest) so every read and write is a coalesced row.
constexpr int BOX_SEG = 64;
__device__ __forceinline__ float box_val(const float* __restrict__ in, size_t base, size_t stride, int q, int len, int mode) {
    if (mode == 0) { if (q < 0 || q >= len) return 0.f; }
    else {                                           // d c b a | a b c d | d c b a
        const int period = 2 * len;
        q %= period; if (q < 0) q += period;
        if (q >= len) q = period - 1 - q;
    }
    return in[base + (size_t)q * stride];
}
__global__ __launch_bounds__(256) void box1d_run_kernel(const float* __restrict__ in, float* __restrict__ out, int X, int Y, int Z,
                                                        int axis, int half, int mode) {
    const int len = axis == 0 ? X : Y;
    const size_t nlines = axis == 0 ? (size_t)Y * Z : (size_t)X * Z;
    const int nseg = (len + BOX_SEG - 1) / BOX_SEG;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nlines * nseg) return;
    const size_t line = t % nlines; const int seg = (int)(t / nlines);
    const size_t stride = axis == 0 ? (size_t)Y * Z : (size_t)Z;
    const size_t base = axis == 0 ? line : (line / Z) * (size_t)Y * Z + (line % Z);
    const int p0 = seg * BOX_SEG, p1 = min(p0 + BOX_SEG, len);
    double acc = 0.0, comp = 0.0;
    for (int d = -half; d <= half; ++d) acc += (double)box_val(in, base, stride, p0 + d, len, mode);
    out[base + (size_t)p0 * stride] = (float)acc;
    for (int p = p0 + 1; p < p1; ++p) {
        const double v = (double)box_val(in, base, stride, p + half, len, mode) - (double)box_val(in, base, stride, p - 1 - half, len, mode);
        const double s = acc + v;                    // TwoSum: s + e == acc + v exactly
        const double bb = s - acc;
        comp += (acc - (s - bb)) + (v - bb);
        acc = s;
        out[base + (size_t)p * stride] = (float)(acc + comp);
    }
}

// a = s / vol ; d = (x - a)^2
__global__ __launch_bounds__(256) void avgdiff_kernel(const float* __restrict__ x, const float* __restrict__ s, float inv_vol, size_t n,
                                                      float* __restrict__ a, float* __restrict__ d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float av = s[i] * inv_vol;
    const float df = x[i] - av;
    a[i] = av; d[i] = df * df;
}

// out = (x - a) / (sqrt(s2 / vol) + noise)
__global__ __launch_bounds__(256) void lcn_final_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ s2,
                                                        float inv_vol, float noise, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (x[i] - a[i]) / (sqrtf(s2[i] * inv_vol) + noise);
}

int select_two_ranks(const void* data, int dtype, size_t n, unsigned long long rank_lo, unsigned long long rank_hi, SelState* st,
                     unsigned int* hist, hipStream_t s) {
    SelState init[2] = {{0u, 0u, rank_lo}, {0u, 0u, rank_hi}};
    HIPCHK(hipMemcpyAsync(st, init, sizeof(init), hipMemcpyHostToDevice, s));
    const int top = dtype == 0 ? 8 : 24;
    const unsigned nblk = (unsigned)((n + 256 * 16 - 1) / (256 * 16) < 2048 ? (n + 256 * 16 - 1) / (256 * 16) : 2048);
    for (int shift = top; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk ? nblk : 1), dim3(256), 0, s, data, dtype, n, shift, st, hist);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(256), 0, s, hist, shift, st);
        LAUNCH_CHECK();
    }
    return CT_OK;
}

}  // namespace

extern "C" {

size_t ct_normalize_workspace_bytes(const int dims[3]) {
    if (!dims || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) return 0;
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    return 4 * align_up(n * sizeof(float), 256) + 4096 + 512;
}

int ct_median(const void* data, int dtype, size_t n, double* median_out, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!data || !median_out || !workspace || n == 0 || (dtype != 0 && dtype != 1)) return CT_EINVAL;
    if (workspace_bytes < 4096) return CT_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    unsigned int* hist = (unsigned int*)ws;                       // 2 x 256 bins
    SelState* st = (SelState*)(ws + 2048);                         // [2]: ranks (n-1)/2 and n/2
    HIPCHK(hipMemsetAsync(hist, 0, 2048, s));
    int rc;
    if ((rc = select_two_ranks(data, dtype, n, (unsigned long long)((n - 1) / 2), (unsigned long long)(n / 2), st, hist, s))) return rc;
    hipLaunchKernelGGL(median_finish_kernel, dim3(1), dim3(1), 0, s, st, st + 1, dtype, median_out);
    LAUNCH_CHECK();
    return CT_OK;
}

int ct_normalize_image(const void* img, int dtype, const int dims[3], double noise_level, const int filter[3], int mode,
                       int subtract_median, float* out, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!img || !dims || !filter || !out || !workspace || (dtype != 0 && dtype != 1) || (mode != 0 && mode != 1)) return CT_EINVAL;
    for (int i = 0; i < 3; ++i) if (dims[i] <= 0 || filter[i] <= 0 || (filter[i] & 1) == 0) return CT_EINVAL;
    if (workspace_bytes < ct_normalize_workspace_bytes(dims)) return CT_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t slab = align_up(n * sizeof(float), 256);
    float* X = (float*)ws; float* A = (float*)(ws + slab); float* T1 = (float*)(ws + 2 * slab); float* T2 = (float*)(ws + 3 * slab);
    unsigned char* tail = ws + 4 * slab;
    double* median = (double*)(tail + 3072);
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (subtract_median) {
        int rc = ct_median(img, dtype, n, median, tail, 4096, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(prep_kernel, dim3(nb), dim3(256), 0, s, img, dtype, n, subtract_median ? median : (const double*)nullptr, X);
    LAUNCH_CHECK();
    const float inv_vol = 1.0f / (float)(filter[0] * filter[1] * filter[2]);
    auto box = [&](const float* src, float* dst, float* tmp) -> int {   // separable: result ends in dst
        const float* cur = src; float* bufs[2] = {dst, tmp}; int w = 0; int passes = 0;
        for (int ax = 0; ax < 3; ++ax) if (filter[ax] > 1) ++passes;
        if (passes == 0) { HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s)); return CT_OK; }
        w = (passes & 1) ? 0 : 1;                         // so that the last pass writes dst
        for (int ax = 0; ax < 3; ++ax) {
            if (filter[ax] == 1) continue;
            if (ax < 2 && filter[ax] > 3) {                    // sliding window along x / y
                const int len = dims[ax];
                const size_t nthreads = (n / len) * (size_t)((len + BOX_SEG - 1) / BOX_SEG);
                hipLaunchKernelGGL(box1d_run_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, cur, bufs[w], dims[0], dims[1],
                                   dims[2], ax, filter[ax] / 2, mode);
            } else
                hipLaunchKernelGGL(box1d_kernel, dim3(nb), dim3(256), 0, s, cur, bufs[w], dims[0], dims[1], dims[2], ax, filter[ax] / 2, mode);
            LAUNCH_CHECK();
            cur = bufs[w]; w ^= 1;
        }
        return CT_OK;
    };
    int rc;
    if ((rc = box(X, T1, T2))) return rc;
    hipLaunchKernelGGL(avgdiff_kernel, dim3(nb), dim3(256), 0, s, X, T1, inv_vol, n, A, T2);
    LAUNCH_CHECK();
    if ((rc = box(T2, T1, out))) return rc;               // `out` doubles as scratch before the final kernel writes it
    hipLaunchKernelGGL(lcn_final_kernel, dim3(nb), dim3(256), 0, s, X, A, T1, inv_vol, (float)noise_level, n, out);
    LAUNCH_CHECK();
    return CT_OK;
}

}  // extern "C"
