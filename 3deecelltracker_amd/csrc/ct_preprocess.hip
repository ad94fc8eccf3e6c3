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
#include <stdlib.h>

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
#ifndef CT_RADIX_PEEL
#define CT_RADIX_PEEL 0
#endif
#ifndef CT_RADIX_BATCH
#define CT_RADIX_BATCH 4
#endif
__global__ __launch_bounds__(256) void radix_hist_kernel(const void* __restrict__ data, int dtype, size_t n, int shift,
                                                         const SelState* __restrict__ st, unsigned int* __restrict__ hist, int nrep) {
    __shared__ unsigned int h[2][256];
    hist += (size_t)(blockIdx.x % nrep) * 512;                 // same-address atomics serialise in L2: 1024 workgroups on one [2][256] table
                                                              // were a 1024-deep chain per bin (20 us); nrep tables make it 1024 / nrep deep
    h[0][threadIdx.x] = 0; h[1][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t p0 = st[0].prefix, p1 = st[1].prefix, mask = st[0].mask;      // the masks are always equal
    const bool same = p0 == p1;
    auto count = [&](bool live, uint32_t k) {
        const int d = (int)((k >> shift) & 0xFF);
        const bool m0 = live && (k & mask) == p0, m1 = live && !same && (k & mask) == p1;
#if CT_RADIX_PEEL > 0
        unsigned long long rem = __ballot(m0);
#pragma unroll 1
        for (int it = 0; it < CT_RADIX_PEEL && rem; ++it) {
            const int lead = __ffsll((long long)rem) - 1;
            const int dl = __shfl(d, lead);
            const unsigned long long grp = __ballot(m0 && d == dl) & rem;
            if ((int)(threadIdx.x & 63) == lead) atomicAdd(&h[0][dl], (unsigned)__popcll(grp));
            rem &= ~grp;
        }
        if ((rem >> (threadIdx.x & 63)) & 1) atomicAdd(&h[0][d], 1u);
#else
        const unsigned long long b0 = __ballot(m0);
        if (b0) {
            const int lead = __ffsll((long long)b0) - 1;
            const int dl = __shfl(d, lead);
            if (__ballot(m0 && d == dl) == b0) { if ((int)(threadIdx.x & 63) == lead) atomicAdd(&h[0][dl], (unsigned)__popcll(b0)); }
            else if (m0) atomicAdd(&h[0][d], 1u);
        }
#endif
        if (m1) atomicAdd(&h[1][d], 1u);
    };
    // 16 bytes per thread and load (8 uint16 or 4 float keys): a 2-byte load per key left the pass at 0.4 TB/s
    const int kpv = dtype == 0 ? 8 : 4;
    const size_t nvec = ((uintptr_t)data & 15) == 0 ? n / kpv : 0;            // (an unaligned view takes the key-by-key loop below)
    const size_t nround = (nvec + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);
    for (size_t rnd = 0; rnd < nround; rnd += CT_RADIX_BATCH) {  // CT_RADIX_BATCH independent loads in flight per thread
        uint4 q[CT_RADIX_BATCH]; bool live[CT_RADIX_BATCH];
#pragma unroll
        for (int b = 0; b < CT_RADIX_BATCH; ++b) {
            const size_t v = ((rnd + b) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
            live[b] = rnd + b < nround && v < nvec;
            q[b] = live[b] ? reinterpret_cast<const uint4*>(data)[v] : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int b = 0; b < CT_RADIX_BATCH; ++b) {
            const uint32_t w[4] = {q[b].x, q[b].y, q[b].z, q[b].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dtype == 0) { count(live[b], w[e] & 0xFFFFu); count(live[b], w[e] >> 16); }
                else count(live[b], w[e] ^ ((w[e] >> 31) ? 0xFFFFFFFFu : 0x80000000u));
            }
        }
    }
    {                                                         // the keys behind the last full vector
        const size_t first = nvec * kpv, rest = n - first;
        const size_t rounds = (rest + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);
        for (size_t rnd = 0; rnd < rounds; ++rnd) {
            const size_t i = first + (rnd * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
            count(i < n, i < n ? key_of(data, dtype, i) : 0u);
        }
    }
    __syncthreads();
    if (h[0][threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[0][threadIdx.x]);
    if (h[1][threadIdx.x]) atomicAdd(&hist[256 + threadIdx.x], h[1][threadIdx.x]);
}

__global__ void radix_pick_kernel(unsigned int* __restrict__ hist, int nrep, int shift, SelState* __restrict__ st) {
    __shared__ unsigned int h[512];
    for (int b = threadIdx.x; b < 512; b += blockDim.x) {
        unsigned int c = 0;
        for (int r = 0; r < nrep; ++r) { c += hist[r * 512 + b]; hist[r * 512 + b] = 0; }      // summed, and ready for the next pass
        h[b] = c;
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                   // one wave: lane l owns bins 4 l .. 4 l + 3 (a serial walk of 256 LDS words took 10 us)
        const int lane = threadIdx.x;
        const bool same = st[0].prefix == st[1].prefix;
        for (int q = 0; q < 2; ++q) {
            const unsigned int* hq = h + ((q == 1 && !same) ? 256 : 0);
            unsigned long long c[4], tot = 0;
            for (int k = 0; k < 4; ++k) { c[k] = hq[4 * lane + k]; tot += c[k]; }
            unsigned long long incl = tot;
            for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            const unsigned long long r = st[q].rank, excl = incl - tot;
            const unsigned long long owner = __ballot(r >= excl && r < incl);
            if (owner ? lane == __ffsll((long long)owner) - 1 : lane == 63) {
                unsigned long long acc = excl; int k = 0;
                for (; k < 4; ++k) { if (acc + c[k] > r) break; acc += c[k]; }
                const int d = k < 4 ? 4 * lane + k : 255;     // (a rank beyond the total cannot occur: the last bin, as the serial walk did)
                st[q].rank = r - acc;
                st[q].prefix |= ((uint32_t)d << shift);
                st[q].mask |= (0xFFu << shift);
            }
        }
    }
}

__global__ void median_finish_kernel(const SelState* __restrict__ lo, const SelState* __restrict__ hi, int dtype,
                                     double* __restrict__ median) {
    *median = 0.5 * (value_of_key(lo->prefix, dtype) + value_of_key(hi->prefix, dtype));      // np.median: mean of the two middles
}

// Source / epilogue plumbing of the box passes.  SRC 0: a float array; SRC 1: the raw image, x = max(img - median, 0) formed on
// the fly (the fp32 copy of x is never written).  EPI 0: store the sum; EPI 1 (last pass of the first box): a = sum / vol ->
// A, (x - a)^2 -> out; EPI 2 (last pass of the second box): (x - A) / (sqrt(sum / vol) + noise) -> out  (preprocess.py:150-167).
struct LcnIO {
    const void* img; int dtype; const double* median;      // the raw image (x is derived from it)
    float* A;                                               // local mean (written by EPI 1, read by EPI 2)
    float inv_vol, noise;
};
__device__ __forceinline__ float lcn_x(const LcnIO& io, size_t i) {
    double v = io.dtype == 0 ? (double)reinterpret_cast<const uint16_t*>(io.img)[i] : (double)reinterpret_cast<const float*>(io.img)[i];
    if (io.median) { v -= *io.median; if (v < 0.0) v = 0.0; }
    return (float)v;
}
template <int SRC> __device__ __forceinline__ float box_src(const float* __restrict__ in, const LcnIO& io, size_t i) {
    if constexpr (SRC == 0) return in[i]; else return lcn_x(io, i);
}
template <int EPI> __device__ __forceinline__ void box_emit(float* __restrict__ out, const LcnIO& io, size_t i, double sum) {
    if constexpr (EPI == 0) out[i] = (float)sum;
    else if constexpr (EPI == 1) { const float av = (float)sum * io.inv_vol; const float df = lcn_x(io, i) - av; io.A[i] = av; out[i] = df * df; }
    else out[i] = (lcn_x(io, i) - io.A[i]) / (sqrtf((float)sum * io.inv_vol) + io.noise);
}
__device__ __forceinline__ int box_fold(int q, int len, int mode) {      // index of tap q, -1 = zero padding
    if (mode == 0) return (q < 0 || q >= len) ? -1 : q;
    const int period = 2 * len;                      // d c b a | a b c d | d c b a
    q %= period; if (q < 0) q += period;
    return q >= len ? period - 1 - q : q;
}

// 1-D centred box sum along `axis` of a [X][Y][Z] array (z fastest); mode 0 = zero padding, 1 = scipy 'reflect'
template <int SRC, int EPI>
__global__ __launch_bounds__(256) void box1d_kernel(const float* __restrict__ in, float* __restrict__ out, int X, int Y, int Z,
                                                    int axis, int half, int mode, LcnIO io) {
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
        const int q = box_fold(pos + d, len, mode);
        if (q >= 0) acc += (double)box_src<SRC>(in, io, base + (size_t)q * stride);
    }
    box_emit<EPI>(out, io, i, acc);
}

// Sliding-window form of box1d_kernel for the x and y axes (stride >= Z): one thread walks a segment of one line, so an
// output costs ~2.7 reads instead of `2 half + 1` (27 by default).  The window sum is carried in double: over a segment of
// BOX_SEG steps it drifts by < BOX_SEG * 2^-53 of the largest partial sum from the directly accumulated double sum -- eleven
// orders of magnitude below the fp32 value that is stored (an error-free TwoSum compensation used to ride along; its dependent
// fp64 chain made the pass latency-bound: 65 us per pass instead of 30); consecutive threads own consecutive lines (z fastest)
// so every read and write is a coalesced row.
#ifndef CT_BOX_SEG
#define CT_BOX_SEG 32
#endif
constexpr int BOX_SEG = CT_BOX_SEG;
template <int SRC>
__device__ __forceinline__ float box_val(const float* __restrict__ in, const LcnIO& io, size_t base, size_t stride, int q, int len, int mode) {
    q = box_fold(q, len, mode);
    return q < 0 ? 0.f : box_src<SRC>(in, io, base + (size_t)q * stride);
}
template <int SRC, int EPI>
__global__ __launch_bounds__(256) void box1d_run_kernel(const float* __restrict__ in, float* __restrict__ out, int X, int Y, int Z,
                                                        int axis, int half, int mode, LcnIO io) {
    const int len = axis == 0 ? X : Y;
    const size_t nlines = axis == 0 ? (size_t)Y * Z : (size_t)X * Z;
    const int nseg = (len + BOX_SEG - 1) / BOX_SEG;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nlines * nseg) return;
    const size_t line = t % nlines; const int seg = (int)(t / nlines);
    const size_t stride = axis == 0 ? (size_t)Y * Z : (size_t)Z;
    const size_t base = axis == 0 ? line : (line / Z) * (size_t)Y * Z + (line % Z);
    const int p0 = seg * BOX_SEG, p1 = min(p0 + BOX_SEG, len);
    double acc = 0.0;
    for (int d = -half; d <= half; ++d) acc += (double)box_val<SRC>(in, io, base, stride, p0 + d, len, mode);
    box_emit<EPI>(out, io, base + (size_t)p0 * stride, acc);
    for (int p = p0 + 1; p < p1; ++p) {
        acc += (double)box_val<SRC>(in, io, base, stride, p + half, len, mode) - (double)box_val<SRC>(in, io, base, stride, p - 1 - half, len, mode);
        box_emit<EPI>(out, io, base + (size_t)p * stride, acc);
    }
}

// Plane form of the same walk: a workgroup first copies one whole plane of lines (every position along `axis` x ZT consecutive z, for one
// value of the other axis) into LDS -- each input element is fetched from memory exactly once, by loads that are all independent of one
// another -- and then the threads walk BOX_SEG-long segments of the lines out of LDS.  box1d_run_kernel fetched 2.8 values per output
// through the caches with two waves per SIMD and sat at 2 TB/s of effective traffic; keeping the window in registers instead made 1.4
// fetches per output and 6 % of the time, i.e. the passes wait on the latency of dependent global loads, not on bytes.  The sums are the
// ones box1d_run_kernel forms (same segments, same order, double accumulator), so the result is bit-identical to it.
constexpr int BOX_PLANE_THREADS = 512, BOX_PLANE_WORDS = 16384;                 // 64 KB of LDS: two planes per CU
template <int SRC, int EPI>
__global__ __launch_bounds__(BOX_PLANE_THREADS) void box1d_plane_kernel(const float* __restrict__ in, float* __restrict__ out, int X, int Y,
                                                                        int Z, int axis, int half, int mode, int ZT, LcnIO io) {
    extern __shared__ float plane[];                          // [len][ZT]
    const int len = axis == 0 ? X : Y;
    const int nzc = (Z + ZT - 1) / ZT;
    const int other = (int)(blockIdx.x / nzc), z0 = (int)(blockIdx.x % nzc) * ZT;
    const int zt = min(ZT, Z - z0);
    const size_t stride = axis == 0 ? (size_t)Y * Z : (size_t)Z;
    const size_t base = (axis == 0 ? (size_t)other * Z : (size_t)other * Y * Z) + z0;
    for (int e = threadIdx.x; e < len * ZT; e += BOX_PLANE_THREADS) {
        const int p = e / ZT, z = e - p * ZT;
        plane[e] = z < zt ? box_src<SRC>(in, io, base + (size_t)p * stride + z) : 0.f;
    }
    __syncthreads();
    const int z = threadIdx.x % ZT, seg = threadIdx.x / ZT;
    const int nseg_blk = BOX_PLANE_THREADS / ZT;
    if (z >= zt || seg >= nseg_blk) return;
    auto tap = [&](int q) { q = box_fold(q, len, mode); return q < 0 ? 0.f : plane[q * ZT + z]; };
    for (int p0 = seg * BOX_SEG; p0 < len; p0 += nseg_blk * BOX_SEG) {
        const int p1 = min(p0 + BOX_SEG, len);
        double acc = 0.0;
        for (int d = -half; d <= half; ++d) acc += (double)tap(p0 + d);
        box_emit<EPI>(out, io, base + (size_t)p0 * stride + z, acc);
        for (int p = p0 + 1; p < p1; ++p) {
            acc += (double)tap(p + half) - (double)tap(p - 1 - half);
            box_emit<EPI>(out, io, base + (size_t)p * stride + z, acc);
        }
    }
}

template <int SRC, int EPI>
int launch_box_pass(bool run, const float* in, float* out, const int dims[3], int ax, int half, int mode, const LcnIO& io, hipStream_t s) {
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    static const bool plane_on = !(getenv("CT_LCN_PLANE") && atoi(getenv("CT_LCN_PLANE")) == 0);
    int zt = dims[ax] <= BOX_PLANE_WORDS ? BOX_PLANE_WORDS / dims[ax] : 0;
    if (zt > dims[2]) zt = dims[2];
    static const int zt_cap = getenv("CT_LCN_ZT") ? atoi(getenv("CT_LCN_ZT")) : 16;   // 16 z per plane: 32 KB of LDS for 512-long lines, so four
    if (zt > zt_cap) zt = zt_cap;                            // planes per CU are in different phases (load / walk + store): 0.206 ms per frame against
                                                             // 0.220 with 32 (all workgroups load, then all store) and 0.267 with 8 (32-byte rows)
    if (run && plane_on && (zt >= 8 || zt == dims[2])) {     // a plane of lines fits in LDS
        const unsigned nblk = (unsigned)(dims[ax == 0 ? 1 : 0] * ((dims[2] + zt - 1) / zt));
        hipLaunchKernelGGL((box1d_plane_kernel<SRC, EPI>), dim3(nblk), dim3(BOX_PLANE_THREADS), (size_t)dims[ax] * zt * sizeof(float), s, in,
                           out, dims[0], dims[1], dims[2], ax, half, mode, zt, io);
    } else if (run) {
        const size_t nthreads = (n / dims[ax]) * (size_t)((dims[ax] + BOX_SEG - 1) / BOX_SEG);
        hipLaunchKernelGGL((box1d_run_kernel<SRC, EPI>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, in, out, dims[0], dims[1],
                           dims[2], ax, half, mode, io);
    } else
        hipLaunchKernelGGL((box1d_kernel<SRC, EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, dims[0], dims[1], dims[2], ax,
                           half, mode, io);
    LAUNCH_CHECK();
    return CT_OK;
}
int box_pass(int src, int epi, bool run, const float* in, float* out, const int dims[3], int ax, int half, int mode, const LcnIO& io, hipStream_t s) {
    switch (src * 3 + epi) {
        case 0: return launch_box_pass<0, 0>(run, in, out, dims, ax, half, mode, io, s);
        case 1: return launch_box_pass<0, 1>(run, in, out, dims, ax, half, mode, io, s);
        case 2: return launch_box_pass<0, 2>(run, in, out, dims, ax, half, mode, io, s);
        case 3: return launch_box_pass<1, 0>(run, in, out, dims, ax, half, mode, io, s);
        case 4: return launch_box_pass<1, 1>(run, in, out, dims, ax, half, mode, io, s);
        default: return launch_box_pass<1, 2>(run, in, out, dims, ax, half, mode, io, s);
    }
}

#ifndef CT_MEDIAN_TABLES
#define CT_MEDIAN_TABLES 16
#endif
constexpr int MEDIAN_TABLES = CT_MEDIAN_TABLES;
constexpr size_t MEDIAN_WS = 256 + MEDIAN_TABLES * 2048 + 256 + 64;   // what ct_median uses at most (+ the median itself behind it)
int select_two_ranks(const void* data, int dtype, size_t n, unsigned long long rank_lo, unsigned long long rank_hi, SelState* st,
                     unsigned int* hist, int nrep, hipStream_t s) {
    SelState init[2] = {{0u, 0u, rank_lo}, {0u, 0u, rank_hi}};
    HIPCHK(hipMemcpyAsync(st, init, sizeof(init), hipMemcpyHostToDevice, s));
    const int top = dtype == 0 ? 8 : 24;
    const unsigned nblk = (unsigned)((n + 256 * 32 - 1) / (256 * 32) < 2048 ? (n + 256 * 32 - 1) / (256 * 32) : 2048);
    for (int shift = top; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk ? nblk : 1), dim3(256), 0, s, data, dtype, n, shift, st, hist, nrep);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(256), 0, s, hist, nrep, shift, st);
        LAUNCH_CHECK();
    }
    return CT_OK;
}

}  // namespace

extern "C" {

size_t ct_normalize_workspace_bytes(const int dims[3]) {
    if (!dims || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) return 0;
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    return 4 * align_up(n * sizeof(float), 256) + MEDIAN_WS + 512;
}

int ct_median(const void* data, int dtype, size_t n, double* median_out, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!data || !median_out || !workspace || n == 0 || (dtype != 0 && dtype != 1)) return CT_EINVAL;
    if (workspace_bytes < 4096) return CT_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t usable = workspace_bytes - (size_t)(ws - (unsigned char*)workspace);
    SelState* st = (SelState*)ws;                                  // [2]: ranks (n-1)/2 and n/2
    unsigned int* hist = (unsigned int*)(ws + 256);                // nrep x [2][256] bins: as many tables as the workspace holds, up to 16
    int nrep = (int)((usable - 256) / 2048);
    nrep = nrep < 1 ? 1 : (nrep > MEDIAN_TABLES ? MEDIAN_TABLES : nrep);
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)nrep * 2048, s));
    int rc;
    if ((rc = select_two_ranks(data, dtype, n, (unsigned long long)((n - 1) / 2), (unsigned long long)(n / 2), st, hist, nrep, s))) return rc;
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
    float* A = (float*)(ws + slab); float* T1 = (float*)(ws + 2 * slab); float* T2 = (float*)(ws + 3 * slab);   // (slab 0: spare)
    unsigned char* tail = ws + 4 * slab;
    double* median = (double*)(tail + MEDIAN_WS - 64);
    if (subtract_median) {
        int rc = ct_median(img, dtype, n, median, tail, MEDIAN_WS - 64, stream);
        if (rc) return rc;
    }
    LcnIO io{img, dtype, subtract_median ? median : (const double*)nullptr, A, 1.0f / (float)(filter[0] * filter[1] * filter[2]), (float)noise_level};
    // One separable box: up to three 1-D passes; the first reads `src` (or x formed from the raw image when src is null), the
    // last one applies the epilogue and writes dst; intermediate sums ping-pong between dst and tmp.
    auto box = [&](const float* src, float* dst, float* tmp, int epi) -> int {
        int axes[3], passes = 0;
        for (int ax = 0; ax < 3; ++ax) if (filter[ax] > 1) axes[passes++] = ax;
        if (passes == 0) axes[passes++] = 2;                  // 1 x 1 x 1: a single pass of width 1 carries source and epilogue
        const float* cur = src; float* bufs[2] = {dst, tmp};
        int w = (passes & 1) ? 0 : 1;                         // so that the last pass writes dst
        for (int k = 0; k < passes; ++k) {
            const int ax = axes[k];
            const bool run = ax < 2 && filter[ax] > 3;        // sliding window along x / y
            const int rc = box_pass(cur == nullptr ? 1 : 0, k == passes - 1 ? epi : 0, run, cur, bufs[w], dims, ax, filter[ax] / 2, mode, io, s);
            if (rc) return rc;
            cur = bufs[w]; w ^= 1;
        }
        return CT_OK;
    };
    int rc;
    if ((rc = box(nullptr, T2, T1, 1))) return rc;            // A = local mean, T2 = (x - A)^2
    if ((rc = box(T2, out, T1, 2))) return rc;                // out = (x - A) / (sqrt(box(T2) / vol) + noise)
    return CT_OK;
}

}  // extern "C"
