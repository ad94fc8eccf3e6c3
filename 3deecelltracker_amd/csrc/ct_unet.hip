// ct_unet.hip -- 3D U-Net sliding-window inference for gfx950 (MI355X), hand-written HIP.
//
// What it replaces (reference, pure Python/Keras): CellTracker/unet3d.py
//   :26-98   unet3_a / unet3_b / unet3_c graph definitions
//   :101-200 conv -> activation -> BatchNorm blocks, max-pool, nearest upsample + concat
//   :203-279 unet3_prediction (reflect pad, patch grid, per-patch model.predict, centre stitch)
//
// Design:
//   * activations live in HBM as fp32 blocks [patch][x][y][C/8][z][8]: one (x,y) column of one
//     8-channel group is 16 voxels x 32 B = 512 contiguous bytes, so every halo-tile load and
//     every epilogue store is a run of full 128-B lines;
//   * a 3x3x3 conv is an implicit GEMM  D[cout][voxel] = sum_k W[cout][k] * A[k][voxel],  k = (tap, cin); a 16-voxel
//     MFMA column is one (x,y) column x 16 z; the A tile (halo included) is staged in LDS once per 8-channel chunk,
//     weights stream L2 -> registers (1 KiB per wave-load, coalesced).  Three arithmetic families share this structure (CT_CONV_MATH,
//     read when a model is created):
//       - conv3_split_kernel<F16 = true> "f16x3" (THE DEFAULT; the first pair of unet3_a's volume path runs in conv_l0l1_fused_kernel): every
//         fp32 operand is split into two fp16 parts after an exact per-patch power-of-two scaling and multiplied on the fp16 matrix pipe
//         (v_mfma_f32_16x16x32_f16, three products hi*hi + hi*lo + lo*hi, fp32 accumulate) -- fp32-faithful results (<= 2.2e-6 per conv
//         block against fp64) at a fifth of the matrix-pipe cycles of the f32-input instruction;
//       - conv3_split_kernel<F16 = false> "bf16x6" (CT_CONV_MATH=bf16x6, round 1's default): three exact bf16 parts, six products on
//         v_mfma_f32_16x16x32_bf16;
//       - conv3_mfma_kernel / _fold / _c8 (CT_CONV_MATH=f32): the exact-fp32 instruction v_mfma_f32_16x16x4_f32
//         (an fmaf chain bit-for-bit; 157 TF peak = the fp32 vector peak);
//     the two non-default families are kept as tested references (tests/test_gpu_unet_modes.py);
//     decoder convs fold the taps that coincide after nearest-neighbour upsampling (12 of 27 per upsampled channel);
//   * bias + LeakyReLU/ReLU + BatchNorm-affine run in the accumulator registers; max-pool, the
//     nearest-upsample + concat of the decoder and the 1x1x1 sigmoid head are fused into the
//     producing / consuming conv so no pooled/upsampled/concatenated tensor is ever materialised
//     separately (pool writes its extra output from registers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <vector>
#include <type_traits>
#include <new>

#include "../../include/ctamd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// CT_ABL: feature ablation of the split conv kernels for the co-residency hazard bisect (scripts/probe/hazard_bisect.sh; results are
// WRONG with any bit set -- the variants only serve as aggressors beside the packed-fp32 victim):
//   1 no MFMAs   2 MFMA operands not read from LDS (zeros)   4 no staging stores to LDS   8 no global tile loads
//   16 no epilogue stores / maxima   32 no SGPR pinning asm   64 no weight loads
//   128 return at the kernel's first statement (resources only)   256 return after the argument fetch / tile decode, before any LDS use
//   2048 epilogue arithmetic kept, global stores (output, pool) skipped   4096 no per-patch maximum (reduction, barrier, atomic)
//   512 no per-wave column tables (no barrier-free LDS write -> read)   1024 the per-patch maxima are not fetched (no threadIdx.y)
//   (conv_l0l1_fused_kernel honours 8, 2048 and 16384 = no L0 MFMAs: profiles/r03_fuse01_ablation.txt)
#ifndef CT_TAP_TABLE
#define CT_TAP_TABLE 1     // tap-slot offsets from a constant table (0: the four-way select per K-block)
#endif
#ifndef CT_ABL
#define CT_ABL 0
#endif
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

static constexpr float kLeakyAlpha = 0.3f;   // keras LeakyReLU() default
static constexpr float kBnEps = 1e-3f;       // keras BatchNormalization() default

// ------------------------------------------------------------------------------------------------
// main conv kernel
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* srcA;      // first  CA input channels (decoder: the low-res tensor, read with >>u)
    const float* srcB;      // next   CB input channels (full resolution)
    int CA, CB;
    int AX, AY, AZ;         // dims of srcA
    int ux, uy, uz;         // log2 upsample factors of srcA (0 or 1)
    int X, Y, Z;            // conv (output) dims
    int nchunks;            // (CA+CB)/8
    const f32x4* wpack;     // [nchunks][14][NT][64] (see pack_conv_weights)
    const float* epi;       // [3][NT*16]: bias | scale | shift
    float* out;             // blocked [P][X][Y][cout/8][Z][8] or null
    float* pool;            // blocked pooled output or null
    int pz;                 // pool factor in z (1 or 2); x,y pooling is always 2x2
    int PX, PY, PZ;         // pooled dims
    const float* head;      // [NT*16] head weights (zero padded) + [1] bias, or null
    float* head_out;        // [P][X][Y][Z]
    int act;                // 0 LeakyReLU(0.3), 1 ReLU
    int tilesX, tilesY, zblocks;
    int tx0, ty0;           // origin (voxels, even) of the launch's tile grid (volume path: only the tiles the centre crops depend on are computed)
    // patches on the volume's far faces keep a shorter crop: their needed / computed extents end earlier (x: patches with grid index
    // i == gx1, y: j == gy1; gx1 < 0: no such patches); workgroups of tiles such a patch does not need return at once
    int p_first, pg_yz, pg_z, gx1, gy1;
    int cx1e, cy1e, nx1e, ny1e;
    uint32_t mdivp[2];      // floor(2^32 / d) for d = pg_yz, pg_z
    int nx0, nx1, ny0, ny1; // outputs some kept voxel depends on: only they enter the tensor's per-patch maximum (the rest of a computed
                            // tile may have been fed from voxels nobody computed)
    float* vol_out;         // head layer of the volume path: probabilities go straight into the stitched volume (centre crop of patch (i, j, k) at
    int vb[3], vc[3], vv[3];   // (i, j, k) * vc, crop origin vb inside the patch, clipped to the volume extents vv) instead of into per-patch maps
    int lowa;               // folded decoder convs: stage the upsampled source at its own resolution (StageGeom LOW; needs ux = uy = 1, uz = 0)
    int sx0, sx1, sy0, sy1; // window of the FULL-RESOLUTION output that some consumer reads (volume path: an encoder conv whose output only feeds a
                            // skip connection is computed in full for its pool, but the decoder reads just the part the centre crops depend on)
    int cout;
    int nt_total;           // cout tiles of 16 in the packed weights (a block computes NT of them)
    int ngroups;            // nt_total / NT  (blocks along the cout dimension; 1 unless Cout > 64)
    // split-fp16 kernels: per-patch absolute maxima of the tensors (uint32 bit patterns of non-negative floats)
    const uint32_t* amaxA;  // srcA's tensor [P] (= amaxB when there is no srcA)
    const uint32_t* amaxB;  // srcB's tensor [P]
    uint32_t* amax_out;     // this conv's output tensor [P] (null for the head layer)
    float wscale_inv;       // 1 / (power-of-two scale applied to the packed fp16 weights)
    int nxcd;               // XCDs the launch stream may use (workgroups are dealt to them round-robin); 0/1 = no remapping
    // split kernels: workgroup id -> tile without integer divisions (set by launch_conv_split from the grid)
    uint32_t xper, xrem;    // grid / nxcd, grid % nxcd
    uint32_t mdiv[5];       // floor(2^32 / d) for d = ngroups, zblocks, tilesY, tilesX, nxcd
#ifdef CT_TRACE
    unsigned long long* trace;   // [workgroups][8] phase timestamps (s_memtime) + hw id, scripts/microbench.py trace
#endif
};

#ifdef CT_TRACE
extern "C" { __attribute__((visibility("default"))) unsigned long long* ct_trace_buf = nullptr;
             __attribute__((visibility("default"))) int ct_trace_layer = -1; }
#define CT_TR(k) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CT_TR(k) do {} while (0)
#endif

namespace {

constexpr int TX = 4, TY = 8;                    // output columns per workgroup (x, y)
constexpr int HX = TX + 2, HY = TY + 2, HZ = 18;  // halo tile (columns, z rows)
constexpr int NF4 = HX * HY * HZ * 2;             // float4 slots of one 8-channel halo tile
constexpr int NSTAGE = (NF4 + 255) / 256;
constexpr int NSLAB = 14;                         // ceil(27 taps * 8 cin / 16 k per slab)
// (Variants measured and removed, see DESIGN.md 4.1: two 8-channel planes per stage, LDS-DMA double-buffered staging,
// XCD-aware workgroup order, a VALU first conv fused with the gather.)

__host__ __device__ constexpr int tap_off(int tap) {   // float offset of tap (dx,dy,dz) in the LDS tile
    return (((tap / 9) * HY + (tap / 3) % 3) * HZ + tap % 3) * 8;
}
__host__ __device__ constexpr int mt_off(int mt) {      // float offset of the wave's m-th column (2 x 4)
    return (((mt >> 2) * HY + (mt & 3)) * HZ) * 8;
}

// One 8-channel halo tile: global -> registers -> (barrier) -> LDS -> (barrier); zero 'same' padding, srcA read at >> u.
__device__ __forceinline__ void stage_halo_tile(const ConvArgs& a, int c0, int p, int x0, int y0, int z0, int tid, float* lds) {
    f32x4 v[NSTAGE];
    const float* src; int CQ, SX, SY, SZ, sux, suy, suz, cq;
    if (c0 < a.CA) { src = a.srcA; CQ = a.CA >> 3; SX = a.AX; SY = a.AY; SZ = a.AZ;
                     sux = a.ux; suy = a.uy; suz = a.uz; cq = c0 >> 3; }
    else           { src = a.srcB; CQ = a.CB >> 3; SX = a.X; SY = a.Y; SZ = a.Z;
                     sux = suy = suz = 0; cq = (c0 - a.CA) >> 3; }
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
        const int f = tid + 256 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (f < NF4) {
            const int col = f / (HZ * 2), w = f - col * (HZ * 2);
            const int hz = w >> 1, half = w & 1;
            const int hx = col / HY, hy = col - hx * HY;
            const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
            if (gx >= 0 && gx < a.X && gy >= 0 && gy < a.Y && gz >= 0 && gz < a.Z) {
                const size_t idx = ((((size_t)(p * SX + (gx >> sux)) * SY + (gy >> suy)) * CQ + cq) * SZ
                                    + (gz >> suz)) * 8 + half * 4;
                v[i] = *reinterpret_cast<const f32x4*>(src + idx);
            }
        }
    }
    __syncthreads();                                          // every wave is done reading the previous tile
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
        const int f = tid + 256 * i;
        if (f < NF4) *reinterpret_cast<f32x4*>(&lds[f * 4]) = v[i];
    }
    __syncthreads();
}

template <int NT>
__global__ __launch_bounds__(256, NT == 2 ? 3 : 2) void conv3_mfma_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NF4 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int cg = b % a.ngroups; b /= a.ngroups;
    const int ntb = cg * NT;                       // first cout tile of this block
    const int zb = b % a.zblocks; b /= a.zblocks;
    const int ty = b % a.tilesY;  b /= a.tilesY;
    const int tx = b % a.tilesX;
    const int p = b / a.tilesX;
    const int x0 = tx * TX, y0 = ty * TY, z0 = zb * 16;

    const int g = lane >> 4, zl = lane & 15;
    const int wx0 = 2 * (wave >> 1), wy0 = 4 * (wave & 1);
    const bool hi = (g >> 1) != 0;
    const int lbase = ((wx0 * HY + wy0) * HZ + zl) * 8 + 4 * (g & 1);

    f32x4 acc[8][NT];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int stage = 0; stage < a.nchunks; ++stage) {
        stage_halo_tile(a, stage * 8, p, x0, y0, z0, tid, lds);
        // ---- 13 full slabs of 16 k-values (2 taps x 8 cin; lane group g owns tap 2s+(g>>1), cin 4(g&1)..+3) and a
        //      half slab for tap 26 (lane group g owns cin 2g, 2g+1 -> two K=4 MFMAs): 27 x 8 = 216 k-values, no padding
        const f32x4* wp = a.wpack + ((size_t)stage * NSLAB * a.nt_total + ntb) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NSLAB - 1; ++s) {
            const int off = lbase + (hi ? tap_off(2 * s + 1) : tap_off(2 * s));
            f32x4 wv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[nt] = wp[(size_t)(s * a.nt_total + nt) * 64];
            f32x4 av[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(&lds[off + mt_off(mt)]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][t], av[mt][t], acc[mt][nt], 0, 0, 0);
        }
        {
            const int off = ((wx0 * HY + wy0) * HZ + zl) * 8 + 2 * g + tap_off(26);
            f32x4 wv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[nt] = wp[(size_t)((NSLAB - 1) * a.nt_total + nt) * 64];
            f32x2 av[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) av[mt] = *reinterpret_cast<const f32x2*>(&lds[off + mt_off(mt)]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][t], av[mt][t], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue: bias -> activation -> BatchNorm affine (BN follows the activation)
    // lane (zl, g) holds, for column mt and n-tile nt, couts 16nt+4g .. +3 of voxel z0+zl
    const int CP = a.nt_total * 16;
    const float alpha = a.act == 0 ? kLeakyAlpha : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cb = 16 * (ntb + nt) + 4 * g;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.epi + cb);
        const f32x4 scale = *reinterpret_cast<const f32x4*>(a.epi + CP + cb);
        const f32x4 shift = *reinterpret_cast<const f32x4*>(a.epi + 2 * CP + cb);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            f32x4 r = acc[mt][nt] + bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = r[e];
                r[e] = (t >= 0.f ? t : t * alpha) * scale[e] + shift[e];
            }
            acc[mt][nt] = r;
        }
    }
    const int z = z0 + zl;
    const int OQ = a.cout >> 3;
    if (a.out) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int x = x0 + wx0 + (mt >> 2), y = y0 + wy0 + (mt & 3);
            if (x < a.X && y < a.Y && z < a.Z) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cb = 16 * (ntb + nt) + 4 * g;
                    if (cb < a.cout) {
                        const size_t idx = ((((size_t)(p * a.X + x) * a.Y + y) * OQ + (cb >> 3)) * a.Z + z) * 8 + (cb & 7);
                        *reinterpret_cast<f32x4*>(a.out + idx) = acc[mt][nt];
                    }
                }
            }
        }
    }
    if (a.pool) {      // MaxPooling3D (2,2,pz): the wave's 2x4 columns are two 2x2 blocks
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int x = x0 + wx0, y = y0 + wy0 + 2 * blk;
            const bool ok = (x + 1 < a.X) && (y + 1 < a.Y) && (z < a.Z);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = fmaxf(fmaxf(acc[2 * blk][nt][e], acc[2 * blk + 1][nt][e]),
                                    fmaxf(acc[4 + 2 * blk][nt][e], acc[5 + 2 * blk][nt][e]));
                    if (a.pz == 2) t = fmaxf(t, __shfl_xor(t, 1));
                    m[e] = t;
                }
                const int cb = 16 * (ntb + nt) + 4 * g;
                const bool zok = (a.pz == 1) || ((zl & 1) == 0 && z + 1 < a.Z);
                if (ok && zok && cb < a.cout) {
                    const int pzc = a.pz == 2 ? (z >> 1) : z;
                    const size_t idx = ((((size_t)(p * a.PX + (x >> 1)) * a.PY + (y >> 1)) * OQ + (cb >> 3)) * a.PZ + pzc) * 8 + (cb & 7);
                    *reinterpret_cast<f32x4*>(a.pool + idx) = m;
                }
            }
        }
    }
    if (a.head) {      // Conv3D(1, 1, activation='sigmoid') fused: dot over channels, then sigmoid
        const float hb = a.head[CP];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            float part = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 hw = *reinterpret_cast<const f32x4*>(a.head + 16 * (ntb + nt) + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) part += acc[mt][nt][e] * hw[e];
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            const int x = x0 + wx0 + (mt >> 2), y = y0 + wy0 + (mt & 3);
            if (g == 0 && x < a.X && y < a.Y && z < a.Z) {
                const float logit = part + hb;
                a.head_out[((size_t)(p * a.X + x) * a.Y + y) * a.Z + z] = 1.f / (1.f + expf(-logit));
            }
        }
    }
}



// ------------------------------------------------------------------------------------------------
// Cout = 8 variant.  With only 8 output channels half of the 16 MFMA rows would be padding, so the rows are
// (x-select, cout): ONE MFMA column now produces the 8 channels of TWO x-adjacent voxels.  Both share a
// 4 x 3 x 3 input footprint (36 taps instead of 2 x 27): 288 k-values = 18 slabs per 8-channel chunk versus
// 2 x 14, i.e. 36 % fewer MFMAs (75 % useful rows instead of 50 %).  The weight tile of row (xs, co) at tap
// (dx', dy, dz) is K[dx' - xs][dy][dz][ci][co] when 0 <= dx' - xs <= 2, else 0 (packed on the host).
// ------------------------------------------------------------------------------------------------
constexpr int NSLAB8 = 18;
__host__ __device__ constexpr int tap_off4(int tap) {   // tap = (dx' * 3 + dy) * 3 + dz, dx' in 0..3
    return (((tap / 9) * HY + (tap / 3) % 3) * HZ + tap % 3) * 8;
}

__global__ __launch_bounds__(256, 2) void conv3_mfma_c8_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NF4 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int zb = b % a.zblocks; b /= a.zblocks;
    const int ty = b % a.tilesY;  b /= a.tilesY;
    const int tx = b % a.tilesX;
    const int p = b / a.tilesX;
    const int x0 = tx * TX, y0 = ty * TY, z0 = zb * 16;
    const int g = lane >> 4, zl = lane & 15;
    const int wx0 = 2 * (wave >> 1), wy0 = 4 * (wave & 1);
    const int lbase = ((wx0 * HY + wy0) * HZ + zl) * 8 + 4 * (g & 1);
    const bool hi = (g >> 1) != 0;

    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        stage_halo_tile(a, chunk * 8, p, x0, y0, z0, tid, lds);

        const f32x4* wp = a.wpack + (size_t)chunk * NSLAB8 * 64 + lane;
#pragma unroll
        for (int s = 0; s < NSLAB8; ++s) {
            const int off = lbase + (hi ? tap_off4(2 * s + 1) : tap_off4(2 * s));
            const f32x4 wv = wp[s * 64];
            f32x4 av[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(&lds[off + mt * HZ * 8]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], av[mt][t], acc[mt], 0, 0, 0);
        }
    }

    // epilogue: lane (zl, g) holds channels 4(g&1)..+3 of the voxel at x = x0 + wx0 + (g>>1), y = y0 + wy0 + mt
    const float alpha = a.act == 0 ? kLeakyAlpha : 0.f;
    const int cb = 4 * (g & 1);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.epi + cb);
    const f32x4 scale = *reinterpret_cast<const f32x4*>(a.epi + 16 + cb);
    const f32x4 shift = *reinterpret_cast<const f32x4*>(a.epi + 32 + cb);
    const int x = x0 + wx0 + (g >> 1), z = z0 + zl;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f32x4 r = acc[mt] + bias;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = r[e];
            r[e] = (t >= 0.f ? t : t * alpha) * scale[e] + shift[e];
        }
        const int y = y0 + wy0 + mt;
        const bool ok = x < a.X && y < a.Y && z < a.Z;
        if (a.out && ok)
            *reinterpret_cast<f32x4*>(a.out + (((size_t)(p * a.X + x) * a.Y + y) * a.Z + z) * 8 + cb) = r;
        if (a.head) {
            const f32x4 hw = *reinterpret_cast<const f32x4*>(a.head + cb);
            float part = r[0] * hw[0] + r[1] * hw[1] + r[2] * hw[2] + r[3] * hw[3];
            part += __shfl_xor(part, 16);                    // the other channel half of the same voxel
            if ((g & 1) == 0 && ok)
                a.head_out[((size_t)(p * a.X + x) * a.Y + y) * a.Z + z] = 1.f / (1.f + expf(-(part + a.head[16])));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Decoder convs: Conv3D over concat([UpSampling3D(2,2,*)(low), skip]).  For the upsampled channels a 3x3 (x, y)
// stencil over nearest-neighbour-doubled data touches only 2 x 2 distinct low-res voxels, which ones depends on the
// output voxel's parity (px, py): taps kx = {0 | 1,2} for even x, {0,1 | 2} for odd x (same in y).  With the weights
// of coinciding taps summed on the host, those channels need 2 x 2 x 3 = 12 taps instead of 27 (6 slabs instead of
// 13.5).  An MFMA shares its weight operand between its 16 columns, so every wave takes the 8 columns of ONE parity
// class of the 4 x 8 tile: wave -> (px, py), column mt -> x = px + 2 (mt >> 2), y = py + 2 (mt & 3).  The LDS tile is
// unchanged (the staging already materialises the upsampled data), the folded taps are a subset of its 27 positions;
// the skip channels run the ordinary 13.5 slabs on the same column mapping.  'same' zero padding is preserved: a
// folded tap that leaves the volume reads the zero halo exactly like the tap it replaces (dims are even).
// ------------------------------------------------------------------------------------------------
constexpr int NFSLAB = 6;                                // 12 folded taps x 8 cin / 16
__host__ __device__ constexpr int ftap_off(int j) {      // folded tap j = (dxi * 2 + dyi) * 3 + dz, relative to (px, py)
    return ((((j / 6) * HY + (j / 3) % 2) * HZ) + j % 3) * 8;
}
__host__ __device__ constexpr int mtf_off(int mt) {      // column mt of a parity class
    return ((2 * (mt >> 2)) * HY + 2 * (mt & 3)) * HZ * 8;
}

template <int NT>
__global__ __launch_bounds__(256, NT == 2 ? 3 : 2) void conv3_mfma_fold_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NF4 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int cg = b % a.ngroups; b /= a.ngroups;
    const int ntb = cg * NT;
    const int zb = b % a.zblocks; b /= a.zblocks;
    const int ty = b % a.tilesY;  b /= a.tilesY;
    const int tx = b % a.tilesX;
    const int p = b / a.tilesX;
    const int x0 = tx * TX, y0 = ty * TY, z0 = zb * 16;

    const int g = lane >> 4, zl = lane & 15;
    const int px = wave >> 1, py = wave & 1;
    const bool hi = (g >> 1) != 0;
    const int cbase = ((px * HY + py) * HZ + zl) * 8;          // column mt = 0 of this wave's parity class
    const int lbase = cbase + 4 * (g & 1);

    f32x4 acc[8][NT];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nA = a.CA >> 3;
    // ---- upsampled channels: 6 slabs of (2 folded taps x 8 cin) with this parity class's summed weights
    for (int stage = 0; stage < nA; ++stage) {
        stage_halo_tile(a, stage * 8, p, x0, y0, z0, tid, lds);
        const int fbase = lbase + (px * HY + py) * HZ * 8;      // folded taps start at halo offset (px, py)
        const f32x4* wp = a.wpack + (((size_t)stage * 4 + (px * 2 + py)) * NFSLAB * a.nt_total + ntb) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NFSLAB; ++s) {
            const int off = fbase + (hi ? ftap_off(2 * s + 1) : ftap_off(2 * s));
            f32x4 wv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[nt] = wp[(size_t)(s * a.nt_total + nt) * 64];
            f32x4 av[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(&lds[off + mtf_off(mt)]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][t], av[mt][t], acc[mt][nt], 0, 0, 0);
        }
    }
    // ---- skip channels: the ordinary 13 slabs + half slab (see conv3_mfma_kernel) on the parity column mapping
    for (int stage = nA; stage < a.nchunks; ++stage) {
        stage_halo_tile(a, stage * 8, p, x0, y0, z0, tid, lds);
        const f32x4* wp = a.wpack + (((size_t)nA * 4 * NFSLAB + (size_t)(stage - nA) * NSLAB) * a.nt_total + ntb) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NSLAB - 1; ++s) {
            const int off = lbase + (hi ? tap_off(2 * s + 1) : tap_off(2 * s));
            f32x4 wv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[nt] = wp[(size_t)(s * a.nt_total + nt) * 64];
            f32x4 av[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(&lds[off + mtf_off(mt)]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][t], av[mt][t], acc[mt][nt], 0, 0, 0);
        }
        {
            const int off = cbase + 2 * g + tap_off(26);
            f32x4 wv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wv[nt] = wp[(size_t)((NSLAB - 1) * a.nt_total + nt) * 64];
            f32x2 av[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) av[mt] = *reinterpret_cast<const f32x2*>(&lds[off + mtf_off(mt)]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][t], av[mt][t], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue (bias -> activation -> BN affine), as in conv3_mfma_kernel but on the parity column mapping
    const int CP = a.nt_total * 16;
    const float alpha = a.act == 0 ? kLeakyAlpha : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cb = 16 * (ntb + nt) + 4 * g;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.epi + cb);
        const f32x4 scale = *reinterpret_cast<const f32x4*>(a.epi + CP + cb);
        const f32x4 shift = *reinterpret_cast<const f32x4*>(a.epi + 2 * CP + cb);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            f32x4 r = acc[mt][nt] + bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = r[e];
                r[e] = (t >= 0.f ? t : t * alpha) * scale[e] + shift[e];
            }
            acc[mt][nt] = r;
        }
    }
    const int z = z0 + zl;
    const int OQ = a.cout >> 3;
    if (a.out) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int x = x0 + px + 2 * (mt >> 2), y = y0 + py + 2 * (mt & 3);
            if (x < a.X && y < a.Y && z < a.Z) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cb = 16 * (ntb + nt) + 4 * g;
                    if (cb < a.cout) {
                        const size_t idx = ((((size_t)(p * a.X + x) * a.Y + y) * OQ + (cb >> 3)) * a.Z + z) * 8 + (cb & 7);
                        *reinterpret_cast<f32x4*>(a.out + idx) = acc[mt][nt];
                    }
                }
            }
        }
    }
    if (a.head) {
        const float hb = a.head[CP];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            float part = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 hw = *reinterpret_cast<const f32x4*>(a.head + 16 * (ntb + nt) + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) part += acc[mt][nt][e] * hw[e];
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            const int x = x0 + px + 2 * (mt >> 2), y = y0 + py + 2 * (mt & 3);
            if (g == 0 && x < a.X && y < a.Y && z < a.Z)
                a.head_out[((size_t)(p * a.X + x) * a.Y + y) * a.Z + z] = 1.f / (1.f + expf(-(part + hb)));
        }
    }
}

// Cout = 8 decoder conv (paired-column kernel + upsample folding).  The pair (x even, x odd) of one MFMA column is
// exactly one low-res x position, so the upsampled channels need the low-res x offsets {-1, 0, +1} (halo dx' = 0, 1, 3)
// -- row (xs = 0) uses {-1: k0, 0: k1 + k2}, row (xs = 1) uses {0: k0 + k1, +1: k2} -- times 2 folded y taps (the wave's
// y parity) times 3 z taps = 18 taps = 9 slabs instead of 18.  wave -> (x pair, py), column mt -> y = py + 2 mt.
constexpr int NFSLAB8 = 9;
__host__ __device__ constexpr int ftap_off8(int j) {       // j = (dxl * 2 + dyi) * 3 + dz, dxl -> halo dx' {0, 1, 3}
    return ((((j / 6) == 2 ? 3 : (j / 6)) * HY + (j / 3) % 2) * HZ + j % 3) * 8;
}

__global__ __launch_bounds__(256, 2) void conv3_mfma_c8_fold_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NF4 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int zb = b % a.zblocks; b /= a.zblocks;
    const int ty = b % a.tilesY;  b /= a.tilesY;
    const int tx = b % a.tilesX;
    const int p = b / a.tilesX;
    const int x0 = tx * TX, y0 = ty * TY, z0 = zb * 16;
    const int g = lane >> 4, zl = lane & 15;
    const int wx0 = 2 * (wave >> 1), py = wave & 1;
    const int lbase = ((wx0 * HY + py) * HZ + zl) * 8 + 4 * (g & 1);
    const int fbase = lbase + py * HZ * 8;
    const bool hi = (g >> 1) != 0;
    constexpr int MTS = 2 * HZ * 8;                        // y stride of the wave's columns

    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nA = a.CA >> 3;
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        stage_halo_tile(a, chunk * 8, p, x0, y0, z0, tid, lds);

        if (chunk < nA) {
            const f32x4* wp = a.wpack + ((size_t)chunk * 2 + py) * NFSLAB8 * 64 + lane;
#pragma unroll
            for (int s = 0; s < NFSLAB8; ++s) {
                const int off = fbase + (hi ? ftap_off8(2 * s + 1) : ftap_off8(2 * s));
                const f32x4 wv = wp[s * 64];
                f32x4 av[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(&lds[off + mt * MTS]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], av[mt][t], acc[mt], 0, 0, 0);
            }
        } else {
            const f32x4* wp = a.wpack + ((size_t)nA * 2 * NFSLAB8 + (size_t)(chunk - nA) * NSLAB8) * 64 + lane;
#pragma unroll
            for (int s = 0; s < NSLAB8; ++s) {
                const int off = lbase + (hi ? tap_off4(2 * s + 1) : tap_off4(2 * s));
                const f32x4 wv = wp[s * 64];
                f32x4 av[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(&lds[off + mt * MTS]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t], av[mt][t], acc[mt], 0, 0, 0);
            }
        }
    }

    // epilogue: lane (zl, g) holds channels 4(g&1)..+3 of the voxel at x = x0 + wx0 + (g>>1), y = y0 + py + 2 mt
    const float alpha = a.act == 0 ? kLeakyAlpha : 0.f;
    const int cb = 4 * (g & 1);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.epi + cb);
    const f32x4 scale = *reinterpret_cast<const f32x4*>(a.epi + 16 + cb);
    const f32x4 shift = *reinterpret_cast<const f32x4*>(a.epi + 32 + cb);
    const int x = x0 + wx0 + (g >> 1), z = z0 + zl;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f32x4 r = acc[mt] + bias;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = r[e];
            r[e] = (t >= 0.f ? t : t * alpha) * scale[e] + shift[e];
        }
        const int y = y0 + py + 2 * mt;
        const bool ok = x < a.X && y < a.Y && z < a.Z;
        if (a.out && ok)
            *reinterpret_cast<f32x4*>(a.out + (((size_t)(p * a.X + x) * a.Y + y) * a.Z + z) * 8 + cb) = r;
        if (a.head) {
            const f32x4 hw = *reinterpret_cast<const f32x4*>(a.head + cb);
            float part = r[0] * hw[0] + r[1] * hw[1] + r[2] * hw[2] + r[3] * hw[3];
            part += __shfl_xor(part, 16);
            if ((g & 1) == 0 && ok)
                a.head_out[((size_t)(p * a.X + x) * a.Y + y) * a.Z + z] = 1.f / (1.f + expf(-(part + a.head[16])));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x6") variants of the conv kernels: fp32 results from the 16x-faster bf16 matrix pipe.
//
// An fp32 value splits EXACTLY into three bf16 numbers by truncation, a = h + m + l (8 + 8 + 8 significant bits;
// bf16 has fp32's exponent range, so no scaling and no overflow/underflow cases).  A product then is
//     a * b = hh + (hm + mh) + (mm + hl + lh) + [ml + lm + ll],
// and dropping the bracket loses at most 2^-23 |ab| -- the size of fp32's own rounding.  bf16 x bf16 products are
// exact in the MFMA's fp32 accumulator, so six v_mfma_f32_16x16x32_bf16 (K = 32 each) replace eight
// v_mfma_f32_16x16x4_f32 (K = 4 each) at about 17 instead of 32 cycles apiece: 2.5x fewer matrix-pipe cycles for the
// same fp32-faithful result (measured error vs an fp64 reference: see DESIGN.md / tests).
//
// Same implicit GEMM and HBM layout as conv3_mfma_kernel; what changes:
//   * staging splits every activation once (4 VALU ops + pack) and writes three bf16 planes [pos][8 ch] (16 B per
//     position per plane, 51.8 KB per workgroup -> still 3 workgroups per CU);
//   * one MFMA consumes K = 32 = 4 taps x 8 channels: lane group g supplies tap 4 kb + g, its 8 channels are one
//     ds_read_b128; 27 taps = 7 K-blocks (the 28th tap slot has zero weights), folded 12 taps = 3, Cout = 8 paired
//     columns 36 taps = 9, folded 18 taps = 5;
//   * weights are split on the host and packed [chunk][(class)][kb][nt][component][lane] x 16 B.
// MODE bits: C8 (paired-column rows = (x-select, cout)), FOLD (decoder conv, parity column mapping, see above).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int KB_STD = 7, KB_FOLD = 3, KB_C8 = 9, KB_C8F = 5;


// ------------------------------------------------------------------------------------------------
// Split-fp16 ("f16x3") family: the same kernels with TWO fp16 components per operand and THREE products.
//
// fp16 carries 11 significant bits: x = hi + lo with hi = the top 11 bits of x and lo = fp16(x - hi) represents x to
// 2^-21 |x| as long as both parts stay inside fp16's exponent range; a * b = hi hi + hi lo + lo hi (+ lo lo ~ 2^-22 |ab|,
// dropped), and fp16 x fp16 products are exact in the MFMA's fp32 accumulator.  Three v_mfma_f32_16x16x32_f16 replace
// the six bf16 products -- half the matrix-pipe cycles again -- for an error of the size of fp32's own rounding.
// fp16's narrow exponent range is handled by exact power-of-two scaling: every conv's epilogue records the per-patch
// absolute maximum of the tensor it writes (one atomicMax per wave), the consumer scales the activations so that this
// maximum lands in [2^13, 2^14) before splitting (values down to 2^-17 of the maximum keep full precision, smaller ones
// an absolute error of 2^-39 of the maximum), the packed weights carry a per-layer power-of-two scale from the host, and
// the epilogue multiplies the accumulators by the exact inverse.  Per-PATCH maxima keep results independent of how
// patches are batched or sharded.
// ------------------------------------------------------------------------------------------------
template <bool F16> struct SplitMath { static constexpr int NC = F16 ? 2 : 3, NP = F16 ? 3 : 6; };

__device__ __forceinline__ void h_split4(const f32x4 v, float s, uint2& h, uint2& l) {
    // hi = the scaled value cut to fp16 (round toward zero: its top 11 significant bits), lo = fp16(x - hi); x - hi is exact in fp32
#if CT_PK
    const f32x2 s2 = f32x2{s, s};
    const f32x2 x01 = f32x2{v[0], v[1]} * s2, x23 = f32x2{v[2], v[3]} * s2;      // (v_pk_mul_f32: the scale is a power of two, the products are exact)
    const float x0 = x01[0], x1 = x01[1], x2 = x23[0], x3 = x23[1];
#else
    const float x0 = v[0] * s, x1 = v[1] * s, x2 = v[2] * s, x3 = v[3] * s;
#endif
    const auto h01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), h23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
    // lo = fp16(v * s - hi) as ONE v_fma_mixlo_f16 / v_fma_mixhi_f16 per value: the fp16 operand is read in place, the difference (exact in fp32, the product
    // being exact) is rounded to nearest and lands in its half of the packed word -- no separate conversion (round 5: v_fma_mix_f32 + v_cvt_pkrtz, lo cut
    // toward zero; nearest halves lo's error)
    f16x2 l01, l23;
    l01[0] = (_Float16)__builtin_fmaf(v[0], s, -(float)h01[0]); l01[1] = (_Float16)__builtin_fmaf(v[1], s, -(float)h01[1]);
    l23[0] = (_Float16)__builtin_fmaf(v[2], s, -(float)h23[0]); l23[1] = (_Float16)__builtin_fmaf(v[3], s, -(float)h23[1]);
    h = uint2{__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
    l = uint2{__builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
}

// exponent k such that m * 2^-k lies in [2^13, 2^14); 0 for m == 0 / inf / nan; clamped so that 2^+-k stay normal floats
__device__ __forceinline__ int amax_exponent(uint32_t mbits) {
    const int e = (int)(mbits >> 23) & 0xff;
    if (e == 0 || e == 0xff) return 0;
    int k = e - 127 - 13;
    return k < -100 ? -100 : (k > 100 ? 100 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

// Every patch's slot has its own 128-B line (same-line atomics serialise in one L2 channel: with the 75 slots of a volume on
// three lines, one atomic per wave cost 1.3 ms per volume) and a workgroup issues ONE no-return atomic (fire and forget; a
// load-and-compare first makes every wave wait for a memory round trip right before it retires: +0.2 ms on the first conv).
constexpr int AMAX_STRIDE = 32;        // uint32 words between the slots of consecutive patches

// wave maximum of non-negative floats by DPP (six 2-cycle VALU ops instead of six ds_bpermute round trips); the result is valid in lane 63
__device__ __forceinline__ float wave_max_nonneg_l63(float m) {
    int v = __builtin_bit_cast(int, m);
#define CT_DPPMAX2(ctrl, rmask) { const int t = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false); v = v > t ? v : t; }
    CT_DPPMAX2(0xB1, 0xf) CT_DPPMAX2(0x4E, 0xf) CT_DPPMAX2(0x141, 0xf) CT_DPPMAX2(0x140, 0xf) CT_DPPMAX2(0x142, 0xa) CT_DPPMAX2(0x143, 0xc)
#undef CT_DPPMAX2
    return __builtin_bit_cast(float, v);
}

// block-wide max of non-negative floats (256 threads = 4 waves) -> one atomicMax on the tensor's per-patch slot
__device__ __forceinline__ void amax_publish(float m, uint32_t* slot, int tid, float* red /* [4] LDS */) {
    // wave maximum by DPP (six 2-cycle VALU ops; __shfl_xor is a chain of six LDS-crossbar round trips): non-negative floats
    // order like their bit patterns, lane 63 ends up with the maximum of all rows
    int v = __builtin_bit_cast(int, m);
#define CT_DPPMAX(ctrl, rmask) { const int t = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false); v = v > t ? v : t; }
    CT_DPPMAX(0xB1, 0xf)      // quad_perm [1,0,3,2]
    CT_DPPMAX(0x4E, 0xf)      // quad_perm [2,3,0,1]
    CT_DPPMAX(0x141, 0xf)     // row_half_mirror
    CT_DPPMAX(0x140, 0xf)     // row_mirror
    CT_DPPMAX(0x142, 0xa)     // row_bcast:15 into rows 1, 3
    CT_DPPMAX(0x143, 0xc)     // row_bcast:31 into rows 2, 3
#undef CT_DPPMAX
    if ((tid & 63) == 63) red[tid >> 6] = __builtin_bit_cast(float, v);
    __syncthreads();
    if (tid == 0)
        (void)__hip_atomic_fetch_max(slot, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
}

// Tile geometry.  Z8 = false: 4 x 8 columns x 16 z (an MFMA column = one (x, y) column).  Z8 = true (levels with Z <= 8,
// e.g. every level of unet3_b): 8 x 8 columns x 8 z, an MFMA column = TWO y-adjacent columns x 8 z (lane bit 3 selects the
// column), so no MFMA lane multiplies padding.
// Y10 (round 5): 4 x 10 columns x 16 z -- every level of unet3_a is a multiple of 10 columns wide (160, 80, 40, 20), the 20 x 20 bottom level is
// NOT a multiple of 8 (three 8-wide tiles cover 24: a fifth of the bottom convs' MFMAs and weight fetches worked on overhang), and a workgroup that
// computes 40 instead of 32 columns per weight fragment fetches a fifth less weight stream for the same outputs.  A wave takes a 2 x 5 block.
template <bool Z8, bool Y10 = false> struct BfGeom {
    static_assert(!(Z8 && Y10), "one geometry at a time");
    static constexpr int TXv = Z8 ? 8 : 4, TYv = Y10 ? 10 : 8, ZB = Z8 ? 8 : 16;
    static constexpr int HXv = TXv + 2, HYv = TYv + 2, HZv = ZB + 2;
    static constexpr int POS = HXv * HYv * HZv;               // halo positions
    static constexpr int PLANE = POS * 16;                    // bytes per component plane
    static constexpr int NF4v = POS * 2, NSTAGEv = (NF4v + 255) / 256;
};
static_assert(BfGeom<false>::HXv == HX && BfGeom<false>::HYv == HY && BfGeom<false>::HZv == HZ, "geometry of the f32 kernels");

// LDS position offset of tap slot t for the four tap sets (0 for the zero-weight padding slots); hy, hz = halo tile strides
__host__ __device__ constexpr int bf_tap_pos(bool c8, bool folded, int t, int hy, int hz, bool low = false) {
    if (folded && low) {                                      // low-resolution tile of the upsampled source: the 2 (x: 3 for a Cout = 8 pair) x 2 distinct voxels are neighbours
        const int nt = c8 ? 18 : 12;
        return t < nt ? ((t / 6) * hy + (t / 3) % 2) * hz + t % 3 : 0;
    }
    if (!folded) {
        const int ntap = c8 ? 36 : 27;
        return t < ntap ? ((t / 9) * hy + (t / 3) % 3) * hz + t % 3 : 0;
    }
    if (!c8) return t < 12 ? ((t / 6) * hy + (t / 3) % 2) * hz + t % 3 : 0;
    return t < 18 ? (((t / 6) == 2 ? 3 : (t / 6)) * hy + (t / 3) % 2) * hz + t % 3 : 0;
}

// position offset of the wave's MFMA column mt (relative to the wave / lane base) for the column mappings of the kernel
__host__ __device__ constexpr int bf_col_pos(bool c8, bool kfold, bool z8, int mt, int hy, int hz, bool low = false, bool y10 = false) {
    if (y10) return ((mt / 5) * hy + (mt % 5)) * hz;                      // plain 2 x 5 block of the wave
    if (low) return c8 ? mt * hz : ((mt >> 2) * hy + (mt & 3)) * hz;       // parity class members are neighbours at half resolution
    if (c8) return mt * (kfold ? 2 : 1) * hz;
    if (!z8) return kfold ? (2 * (mt >> 2) * hy + 2 * (mt & 3)) * hz : ((mt >> 2) * hy + (mt & 3)) * hz;
    return kfold ? (2 * (mt >> 1) * hy + 4 * (mt & 1)) * hz : ((mt >> 2) * hy + 2 * (mt & 3)) * hz;
}

__device__ __forceinline__ void bf_split4(const f32x4 v, uint2& h, uint2& m, uint2& l) {
    uint32_t hb[4], mb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];
        hb[e] = __float_as_uint(x) & 0xffff0000u;
        const float r = x - __uint_as_float(hb[e]);
        mb[e] = __float_as_uint(r) & 0xffff0000u;
        lb[e] = __float_as_uint(r - __uint_as_float(mb[e]));           // <= 8 significant bits left: exact in bf16
    }
    h = uint2{__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u)};
    m = uint2{__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u)};
    l = uint2{__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u)};
}

// Staging of one 8-channel halo tile, global -> registers -> LDS component planes (position p = (hx * HY + hy) * HZ + hz at
// 16 B per plane).  The staging used to be most of the VALU work of the thin layers (and those layers are VALU-issue bound: ~60
// instructions per float4, three quarters of them 64-bit index arithmetic repeated for every channel chunk).  Now:
//   * a thread owns one (z, channel half) of the tile's interior z range and walks the halo columns NCG at a time, so its LDS
//     address and its z offset are per-thread constants and the column part of the address is the same for every thread of a
//     column: it comes from a table in LDS (byte offset of the column in channel group 0 of its patch, -1 = outside the volume =
//     zero padding) built once per workgroup and source tensor;
//   * the two z-halo planes are separate slots, read only when the layer has more than one z block (otherwise they are the 'same'
//     padding: written as zeros with the first chunk and never touched again).
// LOW: the tile of a source that reaches the conv through UpSampling3D(2, 2, 1) is staged at ITS resolution -- (TX / 2 + 2) x (TY / 2 + 2)
// = 4 x 6 columns instead of the 6 x 10 full-resolution halo columns that repeat every voxel up to four times (2.5 x fewer loads, splits
// and LDS stores for half of a decoder conv's chunks); the folded taps then read neighbours (bf_tap_pos / bf_col_pos `low`).
constexpr int LHX = TX / 2 + 2, LHY = TY / 2 + 2;
template <bool Z8, bool LOW = false, bool Y10 = false> struct StageGeom {
    using G = BfGeom<Z8, Y10>;
    static constexpr int ZS = G::ZB * 2;                      // float4 slots of one column's interior: (z, channel half)
    static constexpr int NCG = 256 / ZS;                      // columns per iteration
    static constexpr int NCOLS = LOW ? LHX * LHY : G::HXv * G::HYv;
    static constexpr int NIT = (NCOLS + NCG - 1) / NCG;
    static constexpr int NHS = NCOLS * 4;                     // z-halo slots: (column, plane, channel half)
    static constexpr int NHIT = (NHS + 255) / 256;
};

// Every wave builds its own copy of the table (its lanes write it and read it back: LDS keeps one wave's requests in order, so no
// workgroup barrier stands between the kernel's entry and its first global loads).
template <bool Z8, bool LOW = false, bool Y10 = false>
__device__ __forceinline__ void stage_table(const ConvArgs& a, bool from_a, int x0, int y0, int lane, int* tab) {
    using G = BfGeom<Z8, Y10>;
    const int CQ = from_a ? (a.CA >> 3) : (a.CB >> 3), SY = from_a ? a.AY : a.Y, SZ = from_a ? a.AZ : a.Z;
    const int sux = from_a ? a.ux : 0, suy = from_a ? a.uy : 0;
    if constexpr (LOW) {                                      // columns of the low-resolution source itself (x0, y0 even)
        for (int c = lane; c < LHX * LHY; c += 64) {
            const int lx = c / LHY, ly = c - lx * LHY;
            const int gx = (x0 >> 1) - 1 + lx, gy = (y0 >> 1) - 1 + ly;
            tab[c] = (gx >= 0 && gx < a.AX && gy >= 0 && gy < a.AY) ? ((gx * SY + gy) * CQ * SZ) * 32 : -1;
        }
        return;
    }
#pragma unroll
    for (int c = lane; c < StageGeom<Z8, false, Y10>::NCOLS; c += 64) {
        const int hx = c / G::HYv, hy = c - hx * G::HYv;
        const int gx = x0 - 1 + hx, gy = y0 - 1 + hy;
        tab[c] = (gx >= 0 && gx < a.X && gy >= 0 && gy < a.Y) ? (((gx >> sux) * SY + (gy >> suy)) * CQ * SZ) * 32 : -1;
    }
}

// channel group c0 / 8 of patch p in its source tensor (uniform)
__device__ __forceinline__ const char* stage_base(const ConvArgs& a, int c0, int p) {
    if (c0 < a.CA) return reinterpret_cast<const char*>(a.srcA + ((size_t)p * a.AX * a.AY * (a.CA >> 3) + (c0 >> 3)) * a.AZ * 8);
    return reinterpret_cast<const char*>(a.srcB + ((size_t)p * a.X * a.Y * (a.CB >> 3) + ((c0 - a.CA) >> 3)) * a.Z * 8);
}

template <bool Z8, bool LOW = false, bool Y10 = false>
__device__ __forceinline__ void stage_load(const ConvArgs& a, const char* base, const int* tab, int suz, int z0, int tid, bool with_zhalo,
                                           f32x4 (&v)[StageGeom<Z8, LOW, Y10>::NIT], f32x4 (&vh)[StageGeom<Z8, LOW, Y10>::NHIT]) {
    using S = StageGeom<Z8, LOW, Y10>; using G = BfGeom<Z8, Y10>;
    const int zs = tid % S::ZS, cgp = tid / S::ZS;
    const int gz = z0 + (zs >> 1);
    const int zb = gz < a.Z ? ((gz >> suz) * 8 + (zs & 1) * 4) * 4 : -1;
    int t[S::NIT];
#pragma unroll
    for (int i = 0; i < S::NIT; ++i) {                        // all table reads go out before the first (conditional) global load
        const int c = cgp + S::NCG * i;
        t[i] = ((i + 1) * S::NCG <= S::NCOLS || c < S::NCOLS) ? tab[c] : -1;
    }
#pragma unroll
    for (int i = 0; i < S::NIT; ++i) {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr ((CT_ABL) & 8) { v[i] = f32x4{(float)t[i], 1.f, 2.f, (float)zb}; continue; }
        if ((t[i] | zb) >= 0) v[i] = *reinterpret_cast<const f32x4*>(base + (uint32_t)(t[i] + zb));
    }
    if (with_zhalo) {
#pragma unroll
        for (int i = 0; i < S::NHIT; ++i) {
            const int idx = tid + 256 * i;
            vh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < S::NHS) {
                const int gzh = (idx & 2) ? z0 + G::ZB : z0 - 1;
                const int t = tab[idx >> 2];
                if (t >= 0 && gzh >= 0 && gzh < a.Z && !((CT_ABL) & 8))
                    vh[i] = *reinterpret_cast<const f32x4*>(base + (uint32_t)(t + ((gzh >> suz) * 8 + (idx & 1) * 4) * 4));
            }
        }
    }
}

template <bool Z8, bool F16, bool Y10 = false>
__device__ __forceinline__ void stage_put(const f32x4 v, char* d, float in_scale) {
    using G = BfGeom<Z8, Y10>;
    if constexpr (F16) {
        uint2 h, l;
        h_split4(v, in_scale, h, l);
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + G::PLANE) = l;
    } else {
        uint2 h, m, l;
        bf_split4(v, h, m, l);
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + G::PLANE) = m;
        *reinterpret_cast<uint2*>(d + 2 * G::PLANE) = l;
    }
}

template <bool Z8, bool F16, bool LOW = false, bool Y10 = false>
__device__ __forceinline__ void stage_store(const f32x4 (&v)[StageGeom<Z8, LOW, Y10>::NIT], const f32x4 (&vh)[StageGeom<Z8, LOW, Y10>::NHIT], int tid,
                                            bool with_zhalo, char* lds, float in_scale) {
    using S = StageGeom<Z8, LOW, Y10>; using G = BfGeom<Z8, Y10>;
    const int zs = tid % S::ZS, cgp = tid / S::ZS;
    char* d0 = lds + ((cgp * G::HZv + 1 + (zs >> 1)) * 2 + (zs & 1)) * 8;
#pragma unroll
    for (int i = 0; i < S::NIT; ++i)
        if ((i + 1) * S::NCG <= S::NCOLS || cgp + S::NCG * i < S::NCOLS)
            stage_put<Z8, F16, Y10>(v[i], d0 + i * (S::NCG * G::HZv * 16), in_scale);
    if (with_zhalo) {
#pragma unroll
        for (int i = 0; i < S::NHIT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < S::NHS)
                stage_put<Z8, F16, Y10>(vh[i], lds + (((idx >> 2) * G::HZv + ((idx & 2) ? G::HZv - 1 : 0)) * 2 + (idx & 1)) * 8, in_scale);
        }
    }
}

// byte offset (inside one component plane) of the lane's B fragment for every K-block of a tap set: lane group g supplies tap slot
// 4 kb + g.  Built once per kernel -- inside the K loop the four-way select cost a dozen instructions per K-block.
// The slot positions are a compile-time table in constant memory, one vector load per K-block issued at kernel entry (round 5: written as
// `g == 0 ? t0 : g == 1 ? t1 : ...` the compiler produced ~14 instructions of exec-mask branching per K-block -- VOP3 takes no literals).
template <int KB, bool C8, bool FOLDED, bool Z8, bool LOW, bool Y10> struct TapTable {
    int v[KB * 4];
    constexpr TapTable() : v{} {
        using G = BfGeom<Z8, Y10>;
        constexpr int HYt = LOW ? LHY : G::HYv;
        for (int t = 0; t < KB * 4; ++t) v[t] = bf_tap_pos(C8, FOLDED, t, HYt, G::HZv, LOW) * 16;
    }
};
template <int KB, bool C8, bool FOLDED, bool Z8, bool LOW, bool Y10> __device__ const TapTable<KB, C8, FOLDED, Z8, LOW, Y10> kTapTable{};

template <int KB, bool C8, bool FOLDED, bool Z8, bool LOW = false, bool Y10 = false>
__device__ __forceinline__ void bf_tap_offsets(int lanepos, int g, int (&tapoff)[KB]) {
#if CT_TAP_TABLE
    const int* tab = kTapTable<KB, C8, FOLDED, Z8, LOW, Y10>.v + g;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) tapoff[kb] = lanepos * 16 + tab[4 * kb];
#else
    using G = BfGeom<Z8, Y10>;
    constexpr int HYt = LOW ? LHY : G::HYv;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int t0 = bf_tap_pos(C8, FOLDED, 4 * kb, HYt, G::HZv, LOW), t1 = bf_tap_pos(C8, FOLDED, 4 * kb + 1, HYt, G::HZv, LOW),
                  t2 = bf_tap_pos(C8, FOLDED, 4 * kb + 2, HYt, G::HZv, LOW), t3 = bf_tap_pos(C8, FOLDED, 4 * kb + 3, HYt, G::HZv, LOW);
        tapoff[kb] = (lanepos + (g == 0 ? t0 : (g == 1 ? t1 : (g == 2 ? t2 : t3)))) * 16;
    }
#endif
}

// K-blocks of one chunk: KB blocks of 4 tap slots; weights at wp[((kb * nt_total + nt) * NC + comp) * 64]
#ifndef CT_STORE_FIRST
#define CT_STORE_FIRST 1   // output stores issued before the per-patch maximum is reduced and published (0: after)
#endif
#ifndef CT_PREFETCH
#define CT_PREFETCH 0      // 1: touch the next channel chunk's tile lines (one dummy load per 128-B line) before the current chunk's MFMAs --
#endif                     //    measured 3-12 % SLOWER on every multi-chunk layer (profiles/r05_conv_experiments.txt section 15): experiment, off
#ifndef CT_LB1
#define CT_LB1 4           // workgroups per CU the NT = 1 split-fp16 instantiations are compiled for (<= 128 VGPRs; 3: round 4's bound)
#endif
#ifndef CT_LB2
#define CT_LB2 3           // workgroups per CU the NT = 2 instantiations are compiled for (4: 128 VGPRs -- measured slower, DESIGN App. B.1)
#endif
#ifndef CT_WPF1
#define CT_WPF1 2          // K-blocks of weight fragments in flight ahead of the MFMAs, NT = 1 / 2 / 4
#endif
#ifndef CT_WPF2
#define CT_WPF2 1
#endif
#ifndef CT_WPF4
#define CT_WPF4 0
#endif
template <bool F16, int NT, int NCOL, int KB, bool C8, bool FOLDED, bool KFOLD, bool Z8, bool LOW = false, bool Y10 = false>
__device__ __forceinline__ void bf_chunk_mma(f32x4 (&acc)[NCOL][NT], const char* lds, const int (&tapoff)[KB],
                                             const uint4* wp /* uniform */, uint32_t lane16, int nt_total) {   // (no __restrict__: see the prefetch)
    // A K-block is 12-96 MFMAs (200-1600 cycles); the L2 round trip of its weight fragments is 200+ cycles and nothing else in the
    // wave's stream covers it, so the fragments of the next PFD blocks are requested ahead of this block's MFMAs (registers permitting).
    using G = BfGeom<Z8, Y10>;
    constexpr int NC = SplitMath<F16>::NC, NP = SplitMath<F16>::NP;
    constexpr int PFD0 = NT == 1 ? (CT_WPF1) : (NT == 2 ? (CT_WPF2) : (CT_WPF4));
    constexpr int PFD = PFD0 < KB ? PFD0 : KB - 1;
    u32x4 wbuf[PFD + 1][NT][NC];
    auto wload = [&](int kb, u32x4 (&dst)[NT][NC]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if constexpr ((CT_ABL) & 64) dst[nt][c] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                else dst[nt][c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wp + ((size_t)(kb * nt_total + nt) * NC + c) * 64) + lane16);
            }
    };
#pragma unroll
    for (int k = 0; k < PFD; ++k) wload(k, wbuf[k]);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const char* ab = lds + tapoff[kb];
        if (kb + PFD < KB) wload(kb + PFD, wbuf[(kb + PFD) % (PFD + 1)]);
        u32x4 (&wv)[NT][NC] = wbuf[kb % (PFD + 1)];
        if constexpr (PFD > 0) {
            // keep the prefetch up here: instruction selection and the scheduler sink a plain load to its first use
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int cg = 0; cg < NCOL; cg += 4) {
            u32x4 av[4][NC];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mt = cg + q;
                if (mt >= NCOL) continue;                              // (NCOL = 10: column groups of 4, 4, 2; decided at compile time)
                const int cpos = bf_col_pos(C8, KFOLD, Z8, mt, LOW ? LHY : G::HYv, G::HZv, LOW, Y10);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if constexpr ((CT_ABL) & 2) av[q][c] = u32x4{lane16, lane16, lane16, lane16};
                    else av[q][c] = *reinterpret_cast<const u32x4*>(ab + cpos * 16 + c * G::PLANE);
                }
            }
            // all fragment reads of the column group go out before its first MFMA (left alone, the scheduler issues them just in
            // time to save registers and every pair of MFMAs then eats a full LDS round trip)
            if constexpr (NT == 1) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
            // (weight component, activation component), smallest terms first: bf16 hh, hm, mh, mm, hl, lh; fp16 hh, hl, lh
            constexpr int WI[6] = {0, 0, 1, F16 ? 0 : 1, 0, 2}, AI[6] = {0, 1, 0, F16 ? 0 : 1, 2, 0};
#pragma unroll
            for (int pr = NP - 1; pr >= 0; --pr)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (cg + q >= NCOL) continue;
                        if constexpr ((CT_ABL) & 1) {           // keep the operands alive, one VALU op instead of the MFMA
                            acc[cg + q][nt][0] += __uint_as_float(wv[nt][WI[pr]][0] ^ av[q][AI[pr]][1]);
                        } else
                        if constexpr (F16)
                            acc[cg + q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wv[nt][WI[pr]]),
                                                                                     __builtin_bit_cast(f16x8, av[q][AI[pr]]), acc[cg + q][nt], 0, 0, 0);
                        else
                            acc[cg + q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[nt][WI[pr]]),
                                                                                      __builtin_bit_cast(bf16x8, av[q][AI[pr]]), acc[cg + q][nt], 0, 0, 0);
                    }
        }
    }
}

// max of three / four finite values in one / two instructions.  (fmaxf on values that reached the pool through a two-way branch compiles to a
// canonicalising v_max_f32 x, x, x per operand in front of every maximum: 51 instructions for the 24 maxima of the fused pair's pool.)
__device__ __forceinline__ float max3_nc(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max2_nc(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max4_nc(float a, float b, float c, float d) { return max2_nc(max3_nc(a, b, c), d); }

// Shared epilogue arithmetic of the split kernels: bias -> LeakyReLU / ReLU -> BatchNorm affine on one MFMA result quad, |max| of the four results folded
// into `amax` (v_max3_f32 with |.| modifiers).  Deliberately NOT packed fp32 (CT_PK): v_pk_fma_f32 / v_pk_mul_f32 halve the instruction count and are
// bit-identical, but beside MFMAs they cost more than the two scalar instructions they replace (round 6 A/B: profiles/r06_conv_experiments.txt; the whole
// translation unit is built with -fno-slp-vectorize for that).
#ifndef CT_EPI_DUAL4
#define CT_EPI_DUAL4 0
#endif
#ifndef CT_PK
#define CT_PK 0            // 1: the affine steps of the epilogues and the staging scale as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32) -- measured SLOWER
#endif                     //    beside MFMAs (MI355X_MICROARCH: +22 cycles per v_pk_fma_f32 against two v_fma_f32 in an MFMA gap); A/B switch, off
struct EpiQuad { f32x4 b, s, h; };
__device__ __forceinline__ EpiQuad epi_quad_load(const float* epi_s, int stride, int cb) {
    return EpiQuad{*reinterpret_cast<const f32x4*>(epi_s + cb), *reinterpret_cast<const f32x4*>(epi_s + stride + cb),
                   *reinterpret_cast<const f32x4*>(epi_s + 2 * stride + cb)};
}
// The bias enters the accumulator as its INITIAL value (bias / out_mul: out_mul is a power of two, the quotient exact) and out_mul is folded into the
// BatchNorm scale (LeakyReLU commutes with a positive power-of-two factor): mul, max, fma per value -- one instruction fewer than the round-5 chain
// fma (un-scale + bias), mul, max, fma.  k.s holds scale * out_mul, k.h the shift; equal to that chain up to the position of the bias in the fp32 accumulation.
__device__ __forceinline__ f32x4 epi_quad_apply_c(const f32x4 acc, const float alpha, const EpiQuad& k, float& amax) {
    f32x4 r;
#if CT_PK
    const f32x2 al2 = f32x2{alpha, alpha};
    const f32x2 u01 = f32x2{acc[0], acc[1]} * al2, u23 = f32x2{acc[2], acc[3]} * al2;
    const f32x2 t01 = __builtin_elementwise_fma(f32x2{fmaxf(acc[0], u01[0]), fmaxf(acc[1], u01[1])}, f32x2{k.s[0], k.s[1]}, f32x2{k.h[0], k.h[1]});
    const f32x2 t23 = __builtin_elementwise_fma(f32x2{fmaxf(acc[2], u23[0]), fmaxf(acc[3], u23[1])}, f32x2{k.s[2], k.s[3]}, f32x2{k.h[2], k.h[3]});
    r = f32x4{t01[0], t01[1], t23[0], t23[1]};
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(fmaxf(acc[e], acc[e] * alpha), k.s[e], k.h[e]);      // LeakyReLU / ReLU (0 <= alpha < 1) without a select
#endif
    amax = fmaxf(amax, fmaxf(fabsf(r[0]), fabsf(r[1])));
    amax = fmaxf(amax, fmaxf(fabsf(r[2]), fabsf(r[3])));
    return r;
}
__device__ __forceinline__ float rcp_pow2(float x) { return __uint_as_float(0x7f000000u - __float_as_uint(x)); }   // exact 1 / x for a normal power of two

// hi / lo fp16 split of four values that are ALREADY scaled (the fused pair's L0 outputs: the scale sits in their BatchNorm constants).
// `one` must be an opaque 1.0f (the compiler folds fma(x, 1, c) into an add and the difference then takes a conversion + a subtraction instead of one v_fma_mixlo_f16).
__device__ __forceinline__ void h_split4_scaled(const f32x4 v, float one, uint2& h, uint2& l) {
    const auto h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    f16x2 l01, l23;
    l01[0] = (_Float16)__builtin_fmaf(v[0], one, -(float)h01[0]); l01[1] = (_Float16)__builtin_fmaf(v[1], one, -(float)h01[1]);
    l23[0] = (_Float16)__builtin_fmaf(v[2], one, -(float)h23[0]); l23[1] = (_Float16)__builtin_fmaf(v[3], one, -(float)h23[1]);
    h = uint2{__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
    l = uint2{__builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
}

template <bool F16, int NT, bool C8, bool FOLD, bool Z8, bool Y10 = false>
__global__ __launch_bounds__(256, NT == 4 ? 2 : (NT == 2 ? CT_LB2 : (F16 && !Y10 ? CT_LB1 : 3))) void conv3_split_kernel(const ConvArgs a_in) {
    static_assert(!C8 || NT == 1, "the Cout = 8 kernel has one row tile");
    static_assert(!(C8 && Z8), "Cout = 8 layers sit at the full-resolution level");
    static_assert(!Y10 || (!C8 && !FOLD && !Z8), "4 x 10 tiles: plain layers only");
    using G = BfGeom<Z8, Y10>;
    constexpr int NCOL = C8 ? 4 : (Y10 ? 10 : 8);
    constexpr int HYg = G::HYv, HZg = G::HZv;
    __shared__ __attribute__((aligned(16))) char lds[SplitMath<F16>::NC * G::PLANE];
    __shared__ float amax_red[4];
    __shared__ int coltab[4][2][StageGeom<Z8, false, Y10>::NCOLS];       // per wave: column tables of the two source tensors
    __shared__ __attribute__((aligned(16))) float epi_s[4 * NT * 16];   // this block's bias | scale | shift | head weights
    // All kernel arguments are fetched in one go: left alone, the compiler loads each where it is first used, and the ~20 dependent
    // scalar-load round trips (200-300 cycles apiece with every workgroup hitting the same lines) were a third of a thin layer's
    // workgroup lifetime, all of it in front of the first global load.
    if constexpr ((CT_ABL) & 128) return;
    ConvArgs a = a_in;
    if constexpr (!((CT_ABL) & 32))
    asm volatile("" : "+s"(a.CA), "+s"(a.CB), "+s"(a.AX), "+s"(a.AY), "+s"(a.AZ), "+s"(a.ux), "+s"(a.uy), "+s"(a.uz),
                      "+s"(a.nchunks), "+s"(a.tilesX), "+s"(a.tilesY), "+s"(a.zblocks), "+s"(a.ngroups), "+s"(a.tx0), "+s"(a.ty0),
                      "+s"(a.nxcd), "+s"(a.xper), "+s"(a.xrem), "+s"(a.mdiv[0]), "+s"(a.mdiv[1]), "+s"(a.mdiv[2]), "+s"(a.mdiv[3]), "+s"(a.mdiv[4])
                    : "s"(a_in.X), "s"(a_in.Y), "s"(a_in.Z),       // (X, Y, Z as in-out operands trip the backend: inputs only;
                      "s"(a_in.srcA), "s"(a_in.srcB), "s"(a_in.epi), "s"(a_in.head), "s"(a_in.amaxA), "s"(a_in.amaxB));   // pointers would lose their address space)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    CT_TR(0);
#ifdef CT_TRACE
    if (a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 | (0 << 6) | (31 << 11))) << 32) | __builtin_amdgcn_s_getreg((4 | (0 << 6) | (31 << 11)));
#endif
    // Consecutive workgroup ids go to different XCDs (round-robin), each with its own L2: give every XCD a CONTIGUOUS range of
    // tiles so that neighbouring tiles -- which share their halos -- meet in one L2 instead of fetching them from HBM 8 times
    // (the full-resolution layers moved 1.6-1.8x their algorithmic bytes before this).
    uint32_t b = blockIdx.x;
    // n -> n / d, returns n % d, with the host's floor(2^32 / d): the quotient estimate is short by at most one
    auto divmod = [](uint32_t& n, uint32_t d, uint32_t m) { uint32_t q = __umulhi(n, m), r = n - q * d; if (r >= d) { ++q; r -= d; }
                                                             n = q; return (int)r; };
    {
        const uint32_t xcd = (uint32_t)divmod(b, (uint32_t)a.nxcd, a.mdiv[4]);     // b is now the index inside the XCD's range
        b += xcd * a.xper + (xcd < a.xrem ? xcd : a.xrem);
    }
    const int cg = divmod(b, (uint32_t)a.ngroups, a.mdiv[0]);
    const int ntb = cg * NT;
    const int zb = divmod(b, (uint32_t)a.zblocks, a.mdiv[1]);
    const int ty = divmod(b, (uint32_t)a.tilesY, a.mdiv[2]);
    const int tx = divmod(b, (uint32_t)a.tilesX, a.mdiv[3]);
    const int p = (int)b;
    const int x0 = a.tx0 + tx * G::TXv, y0 = a.ty0 + ty * G::TYv, z0 = zb * G::ZB;
    int nx1 = a.nx1, ny1 = a.ny1;
    int pgi = 0, pgj = 0, pgk = 0;                             // the patch's place in the volume's patch grid (head layer: direct stitch)
    if (a.gx1 >= 0 || a.vol_out) {                             // volume path: is this a patch on a far face of the volume?
        uint32_t pg = (uint32_t)(a.p_first + p);
        const int rem = divmod(pg, (uint32_t)a.pg_yz, a.mdivp[0]);         // pg = i
        uint32_t jq = (uint32_t)rem;
        pgk = divmod(jq, (uint32_t)a.pg_z, a.mdivp[1]);                    // jq = j
        pgi = (int)pg; pgj = (int)jq;
        const bool xe = (int)pg == a.gx1, ye = (int)jq == a.gy1;
        if ((xe && x0 >= a.cx1e) || (ye && y0 >= a.cy1e)) return;          // (uniform, before any barrier) nothing kept depends on this tile
        nx1 = xe ? a.nx1e : nx1; ny1 = ye ? a.ny1e : ny1;
    }
    if constexpr ((CT_ABL) & 256) { if (a.out && x0 == 0x7fffffff) a.out[0] = (float)(y0 + z0 + p + ntb); return; }
    const int g = lane >> 4;
    const int zl = Z8 ? (lane & 7) : (lane & 15);
    const int csel = Z8 ? ((lane >> 3) & 1) : 0;              // Z8: which of the MFMA column's two y-adjacent columns

    // wave -> columns.  plain: 2 x 4 block at (wx, wy) (Z8: 2 x 8 at (2 wave, 0), lane bit 3 = y parity);
    // FOLD: parity class (wx, wy) = (px, py), stride 2 (Z8: lane bit 3 = +2 in y); C8: x pair wx
    const int wx = C8 ? 2 * (wave >> 1) : (FOLD ? (wave >> 1) : (Z8 ? 2 * wave : 2 * (wave >> 1)));
    const int wy = FOLD ? (wave & 1) : (Z8 ? 0 : (Y10 ? 5 : 4) * (wave & 1));
    const int lanepos = (wx * HYg + wy + (FOLD ? 2 : 1) * csel) * HZg + zl;
    // folded taps start at halo offset (px, py) (C8: (0, py))
    const int foldpos = lanepos + (C8 ? wy * HZg : (wx * HYg + wy) * HZg);
    // output coordinates of MFMA column mt (relative to the tile origin)
    auto col_x = [&](int mt) { if constexpr (Y10) return wx + mt / 5; else return C8 ? wx : (FOLD ? (Z8 ? wx + 2 * (mt >> 1) : wx + 2 * (mt >> 2)) : wx + (mt >> 2)); };
    auto col_y = [&](int mt) {
        if constexpr (Y10) return wy + mt % 5; else
        return C8 ? wy + (FOLD ? 2 : 1) * mt
                  : (FOLD ? (Z8 ? wy + 2 * (2 * (mt & 1) + csel) : wy + 2 * (mt & 3)) : (Z8 ? 2 * (mt & 3) + csel : wy + (mt & 3)));
    };

    f32x4 acc[NCOL][NT];
#pragma unroll
    for (int mt = 0; mt < NCOL; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-fp16: one power-of-two scale for all input channels of this patch (the larger of the two source tensors' maxima)
    // (vector loads on purpose -- threadIdx.y is 0, but the compiler cannot know: a scalar load's return is only waitable as
    //  "everything", and that wait would land in front of the first LDS read, i.e. in front of the tile's global loads)
    uint32_t amax_bits = 0, amax_bits_a = 0;
    if constexpr (F16 && !((CT_ABL) & 1024)) {
        amax_bits = a.amaxB[p * AMAX_STRIDE + threadIdx.y];
        amax_bits_a = a.amaxA[p * AMAX_STRIDE + threadIdx.y];     // (the host passes amaxB again when there is no second source)
    }
    // epilogue constants of this block's NT * 16 output channels: fetched now, parked in LDS with the first tile
    constexpr int ECH = NT * 16;
    const int ECP = C8 ? 16 : a.nt_total * 16;                // channel stride of the packed epilogue arrays
    float epi_reg = 0.f;
    {
        const int k = tid / ECH, j = tid - k * ECH;
        if (k < 3) epi_reg = a.epi[k * ECP + ntb * 16 + j];
        else if (k == 3 && a.head) epi_reg = a.head[ntb * 16 + j];
    }
    const float head_bias = a.head ? a.head[ECP + (((CT_ABL) & 1024) ? 0 : threadIdx.y)] : 0.f;
    constexpr int NC = SplitMath<F16>::NC;
    const uint4* wbase = reinterpret_cast<const uint4*>(a.wpack);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    constexpr int KBS = C8 ? KB_C8 : KB_STD;                  // skip / ordinary chunks
    constexpr int KBF = C8 ? KB_C8F : KB_FOLD;                // folded chunks
    constexpr int NCLS = C8 ? 2 : 4;
    const int nA = FOLD ? (a.CA >> 3) : 0;                    // folded chunks
    const int nFromA = a.CA >> 3;                             // chunks read from srcA (the decoder's low-res tensor)
    const int cls = C8 ? wy : (wx * 2 + wy);
    const bool lowa = FOLD && !Z8 && a.lowa != 0;
    if constexpr (!((CT_ABL) & 512)) {
    if (nFromA > 0) { if (lowa) stage_table<Z8, true>(a, true, x0, y0, lane, coltab[wave][0]); else stage_table<Z8, false, Y10>(a, true, x0, y0, lane, coltab[wave][0]); }
    if (nFromA < a.nchunks) stage_table<Z8, false, Y10>(a, false, x0, y0, lane, coltab[wave][1]);
    }
    float in_scale = 1.f, out_mul = 1.f;
#if CT_PREFETCH
    // The next chunk's tile is requested only after this chunk's MFMAs (registers for a real prefetch cost a workgroup per CU, round 2); a dummy
    // load per 128-B line -- one instruction per thread into ONE register that nobody reads -- brings the lines into L2 / MALL meanwhile, so that
    // the real loads find them there.  (inline asm: the compiler must not wait for it; its own vmcnt waits only become conservative)
    float pf_sink = 0.f;
    auto prefetch = [&](int chunk) {
        const bool from_a = chunk < nFromA;
        const bool low = lowa && chunk < nA;
        const int ncols = low ? LHX * LHY : StageGeom<Z8, false, Y10>::NCOLS;
        const int* tab = coltab[wave][from_a ? 0 : 1];
        const int suz = from_a ? a.uz : 0;
        const char* base = stage_base(a, chunk * 8, p);
        const uint32_t zoff = (uint32_t)((z0 >> suz) * 32 + (tid & 3) * 128);
        for (int c = tid >> 2; c < ncols; c += 64) {
            const int t = tab[c];
            if (t >= 0 && ((tid & 3) * 4 + (z0 >> suz)) < (a.Z >> suz)) {
                const uint32_t voff = (uint32_t)t + zoff;
                asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_sink) : "v"(voff), "s"(base) : "memory");
            }
        }
    };
#endif
    auto stage = [&](int chunk, auto low_tag) {
        constexpr bool LOW = decltype(low_tag)::value;
        using SL = StageGeom<Z8, LOW, Y10>;
        f32x4 v[SL::NIT], vh[SL::NHIT];
        const bool from_a = chunk < nFromA;
        const bool zhalo = a.zblocks > 1 || chunk == 0;       // one z block: the z halo is zero padding, written once
        stage_load<Z8, LOW, Y10>(a, stage_base(a, chunk * 8, p), coltab[wave][from_a ? 0 : 1], from_a ? a.uz : 0, z0, tid, zhalo, v, vh);
        if (chunk == 0) CT_TR(1);
        __syncthreads();                                      // every wave is done reading the previous tile
        if (chunk == 0) {
            CT_TR(2);
            if (tid < 4 * ECH) epi_s[tid] = epi_reg;
            if constexpr (F16) {                              // one power-of-two scale for all input channels of this patch
                const int k = amax_exponent(amax_bits_a > amax_bits ? amax_bits_a : amax_bits);
                in_scale = pow2f(-k); out_mul = pow2f(k) * a.wscale_inv;
            }
        }
        if constexpr (!((CT_ABL) & 4)) stage_store<Z8, F16, LOW, Y10>(v, vh, tid, zhalo, lds, in_scale);
        else if (v[0][0] == 12345.678f) lds[tid] = 1;        // (keeps the loads alive)
        if constexpr (LOW) {
            // one z block: the z-halo rows are the 'same' padding, written as zeros with the first chunk only -- for ALL 6 x 10 columns the
            // later full-resolution chunks read, not just the 4 x 6 this chunk staged
            if (chunk == 0 && a.zblocks == 1) {
                for (int idx = tid; idx < StageGeom<Z8, false>::NHS; idx += 256) {
                    char* d = lds + (((idx >> 2) * HZg + ((idx & 2) ? HZg - 1 : 0)) * 2 + (idx & 1)) * 8;
#pragma unroll
                    for (int c = 0; c < NC; ++c) *reinterpret_cast<uint2*>(d + c * G::PLANE) = uint2{0u, 0u};
                }
            }
        }
        __syncthreads();
        if (chunk == 0) {
            CT_TR(3);
            // the bias enters the accumulators as their initial value (bias / out_mul, exact: out_mul is a power of two): the epilogue saves its first fma per
            // value (epi_quad_apply_c).  Read from epi_s, which the barrier above has just published; the first MFMA of every column takes the quad as its C operand.
            const float rom = F16 ? rcp_pow2(out_mul) : 1.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(epi_s + (C8 ? 4 * (g & 1) : 16 * nt + 4 * g)) * rom;
#pragma unroll
                for (int mt = 0; mt < NCOL; ++mt) acc[mt][nt] = bq;
            }
        }
#if CT_PREFETCH
        if constexpr (!(Y10 && NT == 2)) { if (chunk + 1 < a.nchunks) prefetch(chunk + 1); }      // (that instantiation has no register to spare)
#endif
    };
    using NoLow = std::integral_constant<bool, false>;
    if constexpr (FOLD) {
        bool done = false;
        if constexpr (!Z8) {
            if (lowa) {                                       // (uniform) the upsampled chunks from a 4 x 6-column tile of the low-resolution tensor
                int tapoff[KBF];
                const int lowpos = C8 ? ((wx >> 1) * LHY + wy) * HZg + zl : (wx * LHY + wy) * HZg + zl;
                bf_tap_offsets<KBF, C8, true, Z8, true>(lowpos, g, tapoff);
                for (int chunk = 0; chunk < nA; ++chunk) {
                    stage(chunk, std::integral_constant<bool, true>{});
                    const uint4* wp = wbase + (((size_t)chunk * NCLS + cls) * KBF * a.nt_total + ntb) * NC * 64;
                    bf_chunk_mma<F16, NT, NCOL, KBF, C8, true, FOLD, Z8, true>(acc, lds, tapoff, wp, lane16, a.nt_total);
                }
                done = true;
            }
        }
        if (!done) {
            int tapoff[KBF];
            bf_tap_offsets<KBF, C8, true, Z8>(foldpos, g, tapoff);
            for (int chunk = 0; chunk < nA; ++chunk) {
                stage(chunk, NoLow{});
                const uint4* wp = wbase + (((size_t)chunk * NCLS + cls) * KBF * a.nt_total + ntb) * NC * 64;
                bf_chunk_mma<F16, NT, NCOL, KBF, C8, true, FOLD, Z8>(acc, lds, tapoff, wp, lane16, a.nt_total);
            }
        }
    }
    {
        int tapoff[KBS];
        bf_tap_offsets<KBS, C8, false, Z8, false, Y10>(lanepos, g, tapoff);
        for (int chunk = nA; chunk < a.nchunks; ++chunk) {
            stage(chunk, NoLow{});
            const uint4* wp = wbase + (((size_t)nA * NCLS * KBF + (size_t)(chunk - nA) * KBS) * a.nt_total + ntb) * NC * 64;
            bf_chunk_mma<F16, NT, NCOL, KBS, C8, false, FOLD, Z8, false, Y10>(acc, lds, tapoff, wp, lane16, a.nt_total);
        }
    }

    CT_TR(4);
    if constexpr ((CT_ABL) & 16) {
        float sink = 0.f;
#pragma unroll
        for (int mt = 0; mt < NCOL; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sink += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
        if (sink == 12345.678f && a.out) a.out[0] = sink;
        return;
    }
    // ---- epilogue: (un-scale) -> bias -> activation -> BatchNorm affine; stores / fused pool / fused head as in the fp32 kernels
    const float alpha = a.act == 0 ? kLeakyAlpha : 0.f;
    const int z = z0 + zl;
    // head output: per-patch probability map, or (volume path) the patch's centre crop straight into the stitched volume -- what
    // tile_scatter_kernel did in a pass of its own (unet3d.py:246-254)
    auto stitch = [&](int x, int y, int zz, float v) {
        if (a.vol_out) {
            const int lx = x - a.vb[0], ly = y - a.vb[1], lz = zz - a.vb[2];
            if ((unsigned)lx < (unsigned)a.vc[0] && (unsigned)ly < (unsigned)a.vc[1] && (unsigned)lz < (unsigned)a.vc[2]) {
                const int ox = pgi * a.vc[0] + lx, oy = pgj * a.vc[1] + ly, oz = pgk * a.vc[2] + lz;
                if (ox < a.vv[0] && oy < a.vv[1] && oz < a.vv[2]) a.vol_out[((size_t)ox * a.vv[1] + oy) * a.vv[2] + oz] = v;
            }
        } else a.head_out[((size_t)(p * a.X + x) * a.Y + y) * a.Z + zz] = v;
    };
    float vmax = 0.f;                                         // |max| of what this wave writes (split-fp16 consumers scale by it)
    const float al2 = alpha;
    const float om2 = F16 ? out_mul : 1.f;                    // (folded into the BatchNorm scale: epi_quad_apply_c)
    // Wave-uniform shortcuts (round 6): a tile that lies inside the window whose voxels enter the tensor's maximum / inside the stored window / inside the
    // tensor needs no per-column compare chain and no select -- that is every tile but the ones on the windows' borders.
    // (NT = 4: one path only -- CT_EPI_DUAL4: the second copy of a four-row-tile epilogue is ~30 % more code for kernels that are not bound by their epilogue)
    constexpr bool DUAL = NT <= 2 || (CT_EPI_DUAL4);
    const bool tile_needed = DUAL && x0 >= a.nx0 && x0 + G::TXv <= nx1 && y0 >= a.ny0 && y0 + G::TYv <= ny1;
    const bool tile_stored = DUAL && x0 >= a.sx0 && x0 + G::TXv <= a.sx1 && y0 >= a.sy0 && y0 + G::TYv <= a.sy1 && x0 + G::TXv <= a.X && y0 + G::TYv <= a.Y &&
                             z0 + G::ZB <= a.Z;
    if constexpr (C8) {
        // lane (zl, g) holds channels 4(g&1)..+3 of the voxel at x = x0 + wx + (g>>1), y = y0 + col_y(mt)
        const int cb = 4 * (g & 1);
        EpiQuad kq = epi_quad_load(epi_s, ECH, cb);
        kq.s *= om2;
        const int x = x0 + wx + (g >> 1);
        // Cout = 8: one channel octet per voxel; the lane's x (through g >> 1), z and channel half are the lane offset, the column y is scalar
        const size_t patch_bytes8 = (size_t)a.X * a.Y * a.Z * 32;
        const __amdgpu_buffer_rsrc_t rs8 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + (a.out ? (size_t)p * patch_bytes8 : 0), 0,
                                                                             a.out ? (int)patch_bytes8 : 0, 0x00020000);   // (head layer: no tensor, never used)
        const uint32_t lane_off8 = (uint32_t)(((x * a.Y * a.Z + z) * 8 + cb) * 4);
        f32x4 r[4];
        if (tile_needed) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) r[mt] = epi_quad_apply_c(acc[mt][0], al2, kq, vmax);
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float cmax = 0.f;
                r[mt] = epi_quad_apply_c(acc[mt][0], al2, kq, cmax);
                const int y = y0 + col_y(mt);
                if (x >= a.nx0 && x < nx1 && y >= a.ny0 && y < ny1) vmax = fmaxf(vmax, cmax);
            }
        }
        if (a.out && !((CT_ABL) & 2048)) {
            if (tile_stored) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r[mt]), rs8, lane_off8, (y0 + col_y(mt)) * a.Z * 32, 0);
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int y = y0 + col_y(mt);
                    if (x < a.X && y < a.Y && z < a.Z && x >= a.sx0 && x < a.sx1 && y >= a.sy0 && y < a.sy1)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r[mt]), rs8, lane_off8, y * a.Z * 32, 0);
                }
            }
        }
        if (a.head) {
            // Conv3D(1, 1, activation='sigmoid') fused.  The two channel halves of a voxel sit in lanes 16 apart; after the exchange BOTH hold the voxel's
            // sum, so each half finishes two of the four columns (sigmoid + stitch: ~60 vector instructions per column, and this layer is bound by
            // them) instead of one half doing all four while the other idles.  Same arithmetic per voxel as before.
            const f32x4 hw = *reinterpret_cast<const f32x4*>(epi_s + 3 * ECH + cb);
            float part[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                part[mt] = r[mt][0] * hw[0] + r[mt][1] * hw[1] + r[mt][2] * hw[2] + r[mt][3] * hw[3];
                part[mt] += __shfl_xor(part[mt], 16);            // the other channel half of the same voxel
            }
            const bool upper = (g & 1) != 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float pv = upper ? part[2 + k] : part[k];
                const int y = y0 + (upper ? col_y(2 + k) : col_y(k));
                if (x < a.X && y < a.Y && z < a.Z)
                    stitch(x, y, z, 1.f / (1.f + expf(-(pv + head_bias))));
            }
        }
        if constexpr (F16 && !((CT_ABL) & 4096)) { if (a.amax_out) amax_publish(vmax, a.amax_out + p * AMAX_STRIDE, tid, amax_red); }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            EpiQuad kq = epi_quad_load(epi_s, ECH, 16 * nt + 4 * g);
            kq.s *= om2;
            if (!F16 || tile_needed) {                         // (the maximum only serves the split-fp16 consumers)
#pragma unroll
                for (int mt = 0; mt < NCOL; ++mt) acc[mt][nt] = epi_quad_apply_c(acc[mt][nt], al2, kq, vmax);
            } else {
#pragma unroll
                for (int mt = 0; mt < NCOL; ++mt) {
                    float cmax = 0.f;
                    acc[mt][nt] = epi_quad_apply_c(acc[mt][nt], al2, kq, cmax);
                    const int x = x0 + col_x(mt), y = y0 + col_y(mt);
                    if (x >= a.nx0 && x < nx1 && y >= a.ny0 && y < ny1) vmax = fmaxf(vmax, cmax);
                }
            }
        }
#if !CT_STORE_FIRST
        if constexpr (F16 && !((CT_ABL) & 4096)) { if (a.amax_out) amax_publish(vmax, a.amax_out + p * AMAX_STRIDE, tid, amax_red); }
#endif
        const int OQ = a.cout >> 3;
        // stores through a buffer descriptor of this patch's output tensor: scalar column / cout-tile offset + one 32-bit lane offset for
        // all columns (no vector instruction per store; the size_t index expression cost 8 VALU + 10 SALU apiece)
        const int col_bytes = OQ * a.Z * 32;
        if (a.out && !((CT_ABL) & 2048)) {
            const size_t patch_bytes = (size_t)a.X * a.Y * col_bytes;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + (size_t)p * patch_bytes, 0,
                                                                                (int)patch_bytes, 0x00020000);
            const uint32_t lane_off = (uint32_t)((((g >> 1) * a.Z + z) * 8 + 4 * (g & 1)) * 4);
            const int nt_bytes = a.Z * 64;                                 // one 16-channel tile = two channel octets
            constexpr bool ZX = Z8;                                        // (Z8: x / y depend on the lane through csel)
            if (tile_stored && 16 * (ntb + NT) <= a.cout) {                 // (wave-uniform) every lane of every column stores
#pragma unroll
                for (int mt = 0; mt < NCOL; ++mt) {
                    const int x = x0 + col_x(mt), y = y0 + col_y(mt);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (ZX) {
                            const uint32_t voff = lane_off + (uint32_t)((x * a.Y + y) * col_bytes);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][nt]), rs, voff, (ntb + nt) * nt_bytes, 0);
                        } else
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][nt]), rs, lane_off,
                                                                   (x * a.Y + y) * col_bytes + (ntb + nt) * nt_bytes, 0);
                    }
                }
            } else
#pragma unroll
            for (int mt = 0; mt < NCOL; ++mt) {
                const int x = x0 + col_x(mt), y = y0 + col_y(mt);
                if (x >= a.sx0 && x < a.sx1 && y >= a.sy0 && y < a.sy1 && z < a.Z) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (16 * (ntb + nt) + 4 * g < a.cout) {
                            if constexpr (ZX) {
                                const uint32_t voff = lane_off + (uint32_t)((x * a.Y + y) * col_bytes);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][nt]), rs, voff, (ntb + nt) * nt_bytes, 0);
                            } else
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][nt]), rs, lane_off,
                                                                       (x * a.Y + y) * col_bytes + (ntb + nt) * nt_bytes, 0);
                        }
                    }
                }
            }
        }
#if CT_STORE_FIRST
        // (the per-patch maximum -- wave reduction, barrier, one atomic -- after the output stores have been issued: they drain meanwhile)
        if constexpr (F16 && !((CT_ABL) & 4096)) { if (a.amax_out) amax_publish(vmax, a.amax_out + p * AMAX_STRIDE, tid, amax_red); }
#endif
        if constexpr (!FOLD && !Y10) {
            if (a.pool) {      // MaxPooling3D (2,2,pz)
                // !Z8: the wave's 2 x 4 columns are two 2 x 2 blocks {2 blk, 2 blk + 1, 4 + 2 blk, 5 + 2 blk};
                //  Z8: MFMA columns j and j + 4 are x neighbours, the y neighbour sits in lane ^ 8 -> four 2 x 2 blocks
                constexpr int NBLK = Z8 ? 4 : 2;
                const int pcol_bytes = OQ * a.PZ * 32;
                const size_t ppatch_bytes = (size_t)a.PX * a.PY * pcol_bytes;
                const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.pool) + (size_t)p * ppatch_bytes, 0,
                                                                                     (int)ppatch_bytes, 0x00020000);
                const uint32_t plane_off = (uint32_t)((((g >> 1) * a.PZ + (a.pz == 2 ? (z >> 1) : z)) * 8 + 4 * (g & 1)) * 4);
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
                    const int x = x0 + wx, y = y0 + (Z8 ? 2 * blk : wy + 2 * blk);
                    const bool ok = (x + 1 < a.X) && (y + 1 < a.Y) && (z < a.Z) && (!Z8 || csel == 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x4 m;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t;
                            if constexpr (Z8) {
                                t = max2_nc(acc[blk][nt][e], acc[blk + 4][nt][e]);
                                t = max2_nc(t, __shfl_xor(t, 8));
                            } else {
                                t = max4_nc(acc[2 * blk][nt][e], acc[2 * blk + 1][nt][e], acc[4 + 2 * blk][nt][e], acc[5 + 2 * blk][nt][e]);
                            }
                            m[e] = t;
                        }
                        if (a.pz == 2) {                                       // (uniform; one branch per quad) z neighbours sit in adjacent lanes
#pragma unroll
                            for (int e = 0; e < 4; ++e) m[e] = max2_nc(m[e], __shfl_xor(m[e], 1));
                        }
                        const int cb = 16 * (ntb + nt) + 4 * g;
                        const bool zok = (a.pz == 1) || ((zl & 1) == 0 && z + 1 < a.Z);
                        if (ok && zok && cb < a.cout && !((CT_ABL) & 2048)) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), prs, plane_off,
                                                                   ((x >> 1) * a.PY + (y >> 1)) * pcol_bytes + (ntb + nt) * (a.PZ * 64), 0);
                        }
                    }
                }
            }
        }
        if (a.head) {      // Conv3D(1, 1, activation='sigmoid') fused: dot over channels, then sigmoid
            const float hb = head_bias;
#pragma unroll
            for (int mt = 0; mt < NCOL; ++mt) {
                float part = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 hw = *reinterpret_cast<const f32x4*>(epi_s + 3 * ECH + 16 * nt + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) part += acc[mt][nt][e] * hw[e];
                }
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                const int x = x0 + col_x(mt), y = y0 + col_y(mt);
                if (g == 0 && x < a.X && y < a.Y && z < a.Z)
                    stitch(x, y, z, 1.f / (1.f + expf(-(part + hb))));
            }
        }
    }
    CT_TR(5);
#if CT_PREFETCH
    asm volatile("" :: "v"(pf_sink));                        // (the register stays reserved until the last dummy load has landed)
#endif
#ifdef CT_TRACE
    __builtin_amdgcn_s_waitcnt(0); CT_TR(6);
#endif
}

// ------------------------------------------------------------------------------------------------
// first conv (Cin = 1): HBM-bound (AI ~ 12 flop/B), plain VALU, one voxel per thread.
// in [P][X][Y][Z]; w [27][COUT] (scalar loads); out blocked.
// ------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ epi, float* __restrict__ out,
                                                         int P, int X, int Y, int Z, int act, uint32_t* __restrict__ amax_out) {
    __shared__ float amax_red[4];
    const size_t gid0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nvox = (size_t)P * X * Y * Z;
    const bool live = gid0 < nvox;
    const size_t gid = live ? gid0 : nvox - 1;                 // dead lanes recompute the last voxel and do not store
    const int z = (int)(gid % Z); size_t r = gid / Z;
    const int y = (int)(r % Y); r /= Y;
    const int x = (int)(r % X); const int p = (int)(r / X);
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
    const float* base = in + (size_t)p * X * Y * Z;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const int gx = x + dx - 1, gy = y + dy - 1, gz = z + dz - 1;
                float v = 0.f;
                if (gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z)
                    v = base[((size_t)gx * Y + gy) * Z + gz];
                const float* wt = w + ((dx * 3 + dy) * 3 + dz) * COUT;
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v, wt[c], acc[c]);
            }
    const float alpha = act == 0 ? kLeakyAlpha : 0.f;
    constexpr int OQ = COUT / 8;
    float vmax = 0.f;
#pragma unroll
    for (int q = 0; q < OQ; ++q) {
        f32x4 o[2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = q * 8 + e;
            const float t = acc[c] + epi[c];
            o[e >> 2][e & 3] = (t >= 0.f ? t : t * alpha) * epi[COUT + c] + epi[2 * COUT + c];
            vmax = fmaxf(vmax, fabsf(o[e >> 2][e & 3]));
        }
        float* dst = out + ((((size_t)(p * X + x) * Y + y) * OQ + q) * Z + z) * 8;
        if (live) {
            *reinterpret_cast<f32x4*>(dst) = o[0];
            *reinterpret_cast<f32x4*>(dst + 4) = o[1];
        }
    }
    // per-patch |max| of the tensor for the split-fp16 consumers: a block of 256 consecutive voxels lies inside one patch when
    // the patch volume is a multiple of 256 (every reference architecture); otherwise one atomic per lane
    if (amax_out) {
        if (((size_t)X * Y * Z) % 256 == 0) amax_publish(vmax, amax_out + p * AMAX_STRIDE, threadIdx.x, amax_red);
        else (void)__hip_atomic_fetch_max(amax_out + p * AMAX_STRIDE, __float_as_uint(vmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// tiler kernels (unet3d.py:221-255)
// ------------------------------------------------------------------------------------------------
struct TileGeom {
    int vx, vy, vz;      // volume
    int nx, ny, nz;      // net input
    int cx, cy, cz;      // centre size
    int gx, gy, gz;      // grid
    int bx, by, bz;      // 'before' pad = shrink
};

__device__ __forceinline__ int reflect_idx(int i, int n) {   // numpy 'reflect' for any pad width
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    int t = i < 0 ? -i : i;                                   // the index map is even about 0 and periodic with 2 (n - 1) ...
    if (t >= period) t %= period;                             // ... and only pads wider than the volume ever take the division
    return t >= n ? period - t : t;
}

__global__ __launch_bounds__(256) void tile_gather_kernel(const float* __restrict__ vol, TileGeom q, int p_begin,
                                                          int n, float* __restrict__ patches) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)q.nx * q.ny * q.nz;
    if (gid >= per * n) return;
    const int lp = (int)(gid / per); size_t r = gid - (size_t)lp * per;
    const int z = (int)(r % q.nz); r /= q.nz;
    const int y = (int)(r % q.ny); const int x = (int)(r / q.ny);
    const int pg = p_begin + lp;
    const int k = pg % q.gz, j = (pg / q.gz) % q.gy, i = pg / (q.gz * q.gy);
    const int sx = reflect_idx(i * q.cx + x - q.bx, q.vx);
    const int sy = reflect_idx(j * q.cy + y - q.by, q.vy);
    const int sz = reflect_idx(k * q.cz + z - q.bz, q.vz);
    patches[gid] = vol[((size_t)sx * q.vy + sy) * q.vz + sz];
}

// centre crops <-> dense slab [n][cx][cy][cz] (the unit of the multi-GPU gather, SURVEY 8e: every rank contributes the centre
// crops of its own patch range, receives everybody else's and places them).  PACK: volume -> slab (voxels of a crop that lie
// beyond the volume's upper faces are written as 0 so that the slab is fully defined), !PACK: slab -> volume.
template <bool PACK>
__global__ __launch_bounds__(256) void tile_crops_kernel(float* __restrict__ vol, TileGeom q, int p_begin, int n,
                                                         float* __restrict__ crops) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)q.cx * q.cy * q.cz;
    if (gid >= per * n) return;
    const int lp = (int)(gid / per); size_t r = gid - (size_t)lp * per;
    const int z = (int)(r % q.cz); r /= q.cz;
    const int y = (int)(r % q.cy); const int x = (int)(r / q.cy);
    const int pg = p_begin + lp;
    const int k = pg % q.gz, j = (pg / q.gz) % q.gy, i = pg / (q.gz * q.gy);
    const int ox = i * q.cx + x, oy = j * q.cy + y, oz = k * q.cz + z;
    const bool inside = ox < q.vx && oy < q.vy && oz < q.vz;
    const size_t vi = ((size_t)ox * q.vy + oy) * q.vz + oz;
    if (PACK) crops[gid] = inside ? vol[vi] : 0.f;
    else if (inside) vol[vi] = crops[gid];
}

__global__ __launch_bounds__(256) void tile_scatter_kernel(const float* __restrict__ pred, TileGeom q, int p_begin,
                                                           int n, float* __restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)q.cx * q.cy * q.cz;
    if (gid >= per * n) return;
    const int lp = (int)(gid / per); size_t r = gid - (size_t)lp * per;
    const int z = (int)(r % q.cz); r /= q.cz;
    const int y = (int)(r % q.cy); const int x = (int)(r / q.cy);
    const int pg = p_begin + lp;
    const int k = pg % q.gz, j = (pg / q.gz) % q.gy, i = pg / (q.gz * q.gy);
    const int ox = i * q.cx + x, oy = j * q.cy + y, oz = k * q.cz + z;
    if (ox < q.vx && oy < q.vy && oz < q.vz)
        out[((size_t)ox * q.vy + oy) * q.vz + oz] =
            pred[(((size_t)lp * q.nx + q.bx + x) * q.ny + q.by + y) * q.nz + q.bz + z];
}

// ------------------------------------------------------------------------------------------------
// First conv (Cin = 1, Cout = 8) on the matrix cores, fused with the sliding-window gather (volume path): reads the
// reflect-padded volume directly (the patch tensor is never materialised).  Zero 'same' padding applies at the PATCH
// border (patch-local coordinate outside [0, n)), reflect padding at the VOLUME border -- exactly np.pad(..., 'reflect')
// followed by Conv3D(padding='same').  This layer is HBM-bound
// (AI 12 flop/B: it writes 8 channels per input voxel), but a VALU version is issue-bound at 1.3 TB/s.  MFMA rows are
// (x-select, cout) as in conv3_mfma_c8_kernel: one MFMA column = 16 z of TWO x-adjacent voxels sharing a 4 x 3 x 3
// footprint: K = 36 taps = 9 MFMAs (lane group g owns tap 4t + g), the 9 weight values per lane stay in registers,
// the B operand is one ds_read_b32 per MFMA from the 1-channel halo tile.  8 x 8 x 16 outputs per workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int F1X = 8, F1Y = 8, F1Z = 16;
__global__ __launch_bounds__(256, 8) void conv_first_mfma_kernel(const float* __restrict__ vol, TileGeom q, int p_begin,
                                                              const float* __restrict__ wfirst /* [9][64] */,
                                                              const float* __restrict__ epi /* [3][8] */,
                                                              float* __restrict__ out, int act, uint32_t* __restrict__ amax_out) {
    constexpr int HX1 = F1X + 2, HY1 = F1Y + 2, HZ1 = F1Z + 2;
    __shared__ float tile[HX1 * HY1 * HZ1];
    __shared__ int mapx[HX1], mapy[HY1], mapz[HZ1];
    __shared__ float amax_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tilesX = (q.nx + F1X - 1) / F1X, tilesY = (q.ny + F1Y - 1) / F1Y, tilesZ = (q.nz + F1Z - 1) / F1Z;
    int b = blockIdx.x;
    const int tz = b % tilesZ; b /= tilesZ;
    const int ty = b % tilesY; b /= tilesY;
    const int tx = b % tilesX; const int lp = b / tilesX;
    const int pg = p_begin + lp;
    const int pk = pg % q.gz, pj = (pg / q.gz) % q.gy, pi = pg / (q.gz * q.gy);
    const int x0 = tx * F1X, y0 = ty * F1Y, z0 = tz * F1Z;
    if (tid < HX1) { const int l = x0 - 1 + tid; mapx[tid] = (l >= 0 && l < q.nx) ? reflect_idx(pi * q.cx + l - q.bx, q.vx) : -1; }
    else if (tid >= 64 && tid < 64 + HY1) { const int t = tid - 64, l = y0 - 1 + t; mapy[t] = (l >= 0 && l < q.ny) ? reflect_idx(pj * q.cy + l - q.by, q.vy) : -1; }
    else if (tid >= 128 && tid < 128 + HZ1) { const int t = tid - 128, l = z0 - 1 + t; mapz[t] = (l >= 0 && l < q.nz) ? reflect_idx(pk * q.cz + l - q.bz, q.vz) : -1; }
    __syncthreads();
    for (int e = tid; e < HX1 * HY1 * HZ1; e += 256) {
        const int hz = e % HZ1, hy = (e / HZ1) % HY1, hx = e / (HZ1 * HY1);
        const int sx = mapx[hx], sy = mapy[hy], sz = mapz[hz];
        tile[e] = (sx >= 0 && sy >= 0 && sz >= 0) ? vol[((size_t)sx * q.vy + sy) * q.vz + sz] : 0.f;
    }
    const int g = lane >> 4, zl = lane & 15;
    float wreg[9]; int off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wreg[t] = wfirst[t * 64 + lane];
        const int k = 4 * t + g;                                  // tap' = (dx' * 3 + dy) * 3 + dz, dx' in 0..3
        off[t] = ((k / 9) * HY1 + (k / 3) % 3) * HZ1 + k % 3;
    }
    __syncthreads();
    // wave w owns pair-columns: x pair (w >> 1) * 2 + {0, 1}, y = (w & 1) * 4 + {0..3}  -> 8 pairs
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int px0 = (wave >> 1) * 2, py0 = (wave & 1) * 4;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int pxi = px0 + (m >> 2), pyi = py0 + (m & 3);
            const float bv = tile[((2 * pxi) * HY1 + pyi) * HZ1 + zl + off[t]];
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t], bv, acc[m], 0, 0, 0);
        }
    const float alpha = act == 0 ? kLeakyAlpha : 0.f;
    const int cb = 4 * (g & 1);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(epi + cb);
    const f32x4 scale = *reinterpret_cast<const f32x4*>(epi + 8 + cb);
    const f32x4 shift = *reinterpret_cast<const f32x4*>(epi + 16 + cb);
    const int z = z0 + zl;
    float vmax = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = x0 + 2 * (px0 + (m >> 2)) + (g >> 1), y = y0 + py0 + (m & 3);
        if (x >= q.nx || y >= q.ny || z >= q.nz) continue;
        f32x4 r = acc[m] + bias;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = r[e]; r[e] = fmaxf(t, t * alpha) * scale[e] + shift[e]; vmax = fmaxf(vmax, fabsf(r[e])); }
        *reinterpret_cast<f32x4*>(out + (((size_t)(lp * q.nx + x) * q.ny + y) * q.nz + z) * 8 + cb) = r;
    }
    if (amax_out) amax_publish(vmax, amax_out + lp * AMAX_STRIDE, tid, amax_red);
}

// ------------------------------------------------------------------------------------------------
// First conv on the fp16 matrix pipe (split-fp16 family).  Same tile, gather and row mapping as conv_first_mfma_kernel; what
// changes is the arithmetic: the 1-channel halo tile is converted once to packed (hi, lo) fp16 pairs (4 B per voxel, the size
// of the fp32 value) with a power-of-two scale taken from the TILE's own maximum (Cin = 1: a single chunk, so every workgroup
// can scale by itself), and the three products hi*hi + lo*hi + hi*lo of a tap sit in three K slots of ONE MFMA:
//   K = 32 = 4 lane groups x 2 taps x [a_hi, a_lo, a_hi, 0] . [w_hi, w_hi, w_lo, 0]
// so 36 taps = 5 v_mfma_f32_16x16x32_f16 (80 cycles) instead of 9 v_mfma_f32_16x16x4_f32 (288 cycles) per 32 voxels, fed by two
// ds_read_b32 + two v_and per MFMA.  The layer is HBM-bound; this takes the matrix phase out of its critical path.
// ------------------------------------------------------------------------------------------------
template <int OCC>
__global__ __launch_bounds__(256, OCC) void conv_first_f16_kernel(const float* __restrict__ vol, TileGeom q, int p_begin,
                                                             const u32x4* __restrict__ wf16 /* [5][64] */, float wscale_inv,
                                                             const float* __restrict__ epi /* [3][8] */,
                                                             float* __restrict__ out, int act, uint32_t* __restrict__ amax_out) {
    constexpr int HX1 = F1X + 2, HY1 = F1Y + 2, HZ1 = F1Z + 2, NV = HX1 * HY1 * HZ1;
    __shared__ uint32_t tile[NV];
    __shared__ int mapx[HX1], mapy[HY1], mapz[HZ1];
    __shared__ float amax_red[4], tmax_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tilesX = (q.nx + F1X - 1) / F1X, tilesY = (q.ny + F1Y - 1) / F1Y, tilesZ = (q.nz + F1Z - 1) / F1Z;
    int b = blockIdx.x;
    const int tz = b % tilesZ; b /= tilesZ;
    const int ty = b % tilesY; b /= tilesY;
    const int tx = b % tilesX; const int lp = b / tilesX;
    const int pg = p_begin + lp;
    const int pk = pg % q.gz, pj = (pg / q.gz) % q.gy, pi = pg / (q.gz * q.gy);
    const int x0 = tx * F1X, y0 = ty * F1Y, z0 = tz * F1Z;
    if (tid < HX1) { const int l = x0 - 1 + tid; mapx[tid] = (l >= 0 && l < q.nx) ? reflect_idx(pi * q.cx + l - q.bx, q.vx) : -1; }
    else if (tid >= 64 && tid < 64 + HY1) { const int t = tid - 64, l = y0 - 1 + t; mapy[t] = (l >= 0 && l < q.ny) ? reflect_idx(pj * q.cy + l - q.by, q.vy) : -1; }
    else if (tid >= 128 && tid < 128 + HZ1) { const int t = tid - 128, l = z0 - 1 + t; mapz[t] = (l >= 0 && l < q.nz) ? reflect_idx(pk * q.cz + l - q.bz, q.vz) : -1; }
    const int g = lane >> 4, zl = lane & 15;
    __syncthreads();
    // gather: thread t < 2 * HX1 * HY1 takes half a z column (HZ1 / 2 values) of halo column t >> 1 -- the (x, y) reflect
    // look-ups and the row base address are computed once per thread, each element costs one add and one load
    constexpr int HALF = HZ1 / 2;
    static_assert(HZ1 % 2 == 0 && 2 * HX1 * HY1 <= 256, "gather mapping");
    float vals[HALF]; float tmax = 0.f;
    const int gcol = tid >> 1, gz0 = (tid & 1) * HALF;
    const bool gact = tid < 2 * HX1 * HY1;
    {
        const int hx = gcol / HY1, hy = gcol - hx * HY1;
        const int sx = gact ? mapx[hx] : -1, sy = gact ? mapy[hy] : -1;
        const float* rowp = vol + ((size_t)(sx < 0 ? 0 : sx) * q.vy + (sy < 0 ? 0 : sy)) * q.vz;
        const bool rowok = sx >= 0 && sy >= 0;
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const int sz = mapz[gz0 + i];
            const float v = (rowok && sz >= 0) ? rowp[sz] : 0.f;
            vals[i] = v; tmax = fmaxf(tmax, fabsf(v));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
    if (lane == 0) tmax_red[wave] = tmax;
    __syncthreads();
    const int kexp = amax_exponent(__float_as_uint(fmaxf(fmaxf(tmax_red[0], tmax_red[1]), fmaxf(tmax_red[2], tmax_red[3]))));
    const float in_scale = pow2f(-kexp), out_mul = pow2f(kexp) * wscale_inv;
    if (gact) {
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const float x = vals[i] * in_scale;
            const float hi = __uint_as_float(__float_as_uint(x) & 0xffffe000u);
            tile[gcol * HZ1 + gz0 + i] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(hi, x - hi));   // [15:0] hi, [31:16] lo
        }
    }
    // tap' = (dx' * 3 + dy) * 3 + dz (dx' in 0..3) of MFMA j, lane group g, sub-slot s = 8 j + 2 g + s; slots beyond 36 carry zero weights
    __syncthreads();
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int px0 = (wave >> 1) * 2, py0 = (wave & 1) * 4;
    const int wbase = ((2 * px0) * HY1 + py0) * HZ1 + zl;
    u32x4 wcur = wf16[lane];                                        // weights stream from L1/L2 one K-block ahead (20 VGPRs if all were held)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const u32x4 wj = wcur;
        if (j + 1 < 5) wcur = wf16[(j + 1) * 64 + lane];
        const int k0 = 8 * j + 2 * g, k1 = k0 + 1;
        const int o0 = wbase + (k0 < 36 ? ((k0 / 9) * HY1 + (k0 / 3) % 3) * HZ1 + k0 % 3 : 0);
        const int o1 = wbase + (k1 < 36 ? ((k1 / 9) * HY1 + (k1 / 3) % 3) * HZ1 + k1 % 3 : 0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            constexpr int dummy = 0; (void)dummy;
            const int cm = ((2 * (m >> 2)) * HY1 + (m & 3)) * HZ1;          // compile-time column offset
            const uint32_t a0 = tile[o0 + cm], a1 = tile[o1 + cm];
            const u32x4 av = u32x4{a0, a0 & 0xffffu, a1, a1 & 0xffffu};
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wj), __builtin_bit_cast(f16x8, av), acc[m], 0, 0, 0);
        }
    }
    const float alpha = act == 0 ? kLeakyAlpha : 0.f;
    const int cb = 4 * (g & 1);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(epi + cb);
    const f32x4 scale = *reinterpret_cast<const f32x4*>(epi + 8 + cb);
    const f32x4 shift = *reinterpret_cast<const f32x4*>(epi + 16 + cb);
    const int z = z0 + zl;
    float vmax = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = x0 + 2 * (px0 + (m >> 2)) + (g >> 1), y = y0 + py0 + (m & 3);
        if (x >= q.nx || y >= q.ny || z >= q.nz) continue;
        f32x4 r = acc[m] * out_mul + bias;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = r[e]; r[e] = fmaxf(t, t * alpha) * scale[e] + shift[e]; vmax = fmaxf(vmax, fabsf(r[e])); }
        *reinterpret_cast<f32x4*>(out + (((size_t)(lp * q.nx + x) * q.ny + y) * q.nz + z) * 8 + cb) = r;
    }
    if (amax_out) amax_publish(vmax, amax_out + lp * AMAX_STRIDE, tid, amax_red);
}

// ------------------------------------------------------------------------------------------------
// First conv fused INTO the second (volume path of unet3_a, split-fp16 family): L0 (1 -> 8) is evaluated on the 6 x 10 x 16 halo tile of
// an L1 (8 -> 16) workgroup straight from the reflect-padded volume and handed to L1's MFMA phase through LDS -- the 8-channel tensor
// between them (0.98 GB per 512 x 512 x 32 volume, written once and read once) never exists.  L0 is cheap enough to recompute on the halo
// (1.9 x its outputs, 150 of the tile's 820 MFMAs); L1 keeps its tile, weights, pool and epilogue.  Cin(L1) = 8 is a single chunk, so
// one power-of-two scale per TILE (the maximum of the L0 values the tile holds) serves the fp16 split -- as conv_first_f16_kernel does for
// its 1-channel input.  Values differ from the two-kernel path only through those scale exponents (both are exact splits of fp32 data).
// Preconditions (run_network): one z block (Z = 16), Cout(L1) = 16, plain (not folded, not 8 x 8 x 8) tiles, whole-patch region.
// ------------------------------------------------------------------------------------------------
// bound_a, bound_b: |L0 output| <= bound_a * max|input| + bound_b for every channel (host: max_c |scale_c| * sum |w_c|, max_c (|scale_c| |bias_c| + |shift_c|))
struct FirstArgs { const float* vol; TileGeom q; int p_begin; const u32x4* wf16; float wscale_inv; const float* epi; float bound_a, bound_b; };

// (dy, dz) combination `idx` (0 .. 8; 9 = the zero-weight pad slot) of the fused kernel's L0 tap order -> entry offset inside the input tile
// (the pad slot reads the entry after (2, 2): one ds_read2_b64 with adjacent offsets; for the tile's last column that is the zeroed entry behind the tile)
__host__ __device__ constexpr int l0_tap_entry(int idx, int iz) { return idx < 9 ? (idx / 3) * iz + idx % 3 : 2 * iz + 3; }

// Round 6: the kernel was VALU-issue bound (1307 vector + 764 scalar instructions around 208 MFMAs per wave and tile; profiles/r05_conv_experiments.txt
// section 4, round-5 verdict).  What changed, in the order of the phases:
//   * L0's K order is (lane group g = x offset dx' of the output PAIR, K-block j = (dy, dz) combinations 2j, 2j + 1): the lane part of a B-fragment
//     address is dx' alone, the tap part is a compile-time immediate -- one ds_read2_b64 per MFMA and no vector instruction (before: two ds_read_b32, two
//     adds, two ands, and a wave-uniform branch around every MFMA that kept the LDS reads of a column from overlapping the previous MFMA).  The input
//     tile holds 8 B per voxel {hi | lo << 16, hi}, the second word precomputed by the gather;
//   * every wave runs eight pair columns (the two spare slots of waves 2 and 3 recompute pair 29 and write the same values): straight-line code;
//   * tiles whose L0 halo lies inside the patch (85 % of them) skip the zero-padding selects, tiles inside the needed / stored windows skip the per-column
//     compare chains (one wave-uniform test each);
//   * the staging offsets of L0's outputs are scalar column + lane constant;
//   * L1's input scale comes from a BOUND of L0's outputs (bound_a * max|input tile| + bound_b, known before L0 runs) instead of their measured maximum: no second
//     reduction, and the scale folds into L0's BatchNorm constants, so the staged values leave the epilogue ready to split (the bound exceeds the true maximum by
//     the slack of the triangle inequality, a few bits of the 2^17 of head-room the split has below the maximum);
//   * both biases enter their accumulators as initial values and out_mul folds into the BatchNorm scale (epi_quad_apply_c): three instructions per value.
__global__ __launch_bounds__(256, 3) void conv_l0l1_fused_kernel(const ConvArgs a_in, const FirstArgs f_in) {
    using G = BfGeom<false>;
    constexpr int HYg = G::HYv, HZg = G::HZv;                                 // L1 halo tile: 6 x 10 columns x 18 z
    constexpr int IX = G::HXv + 2, IY = G::HYv + 2, IZ = G::ZB + 2;           // L0's input tile: 8 x 12 x 18
    constexpr int NPAIR = (G::HXv / 2) * G::HYv;                              // 30 x-pairs of L0 outputs
    __shared__ __attribute__((aligned(16))) char lds[2 * G::PLANE];
    // L0's packed input tile (8 B per voxel) lives in the first 13.8 KB of L1's planes: nobody writes the planes before every wave has finished its L0
    // MFMAs (the barrier of the output-maximum reduction), and 35 KB instead of 49 keep FOUR workgroups on a CU like the plain L1 kernel
    uint2* const itile = reinterpret_cast<uint2*>(lds);
    static_assert((IX * IY * IZ + 1) * 8 <= G::PLANE, "input tile (+ the pad entry) must fit into the first plane");
    __shared__ int mapx[IX], mapy[IY], mapz[IZ];
    __shared__ float amax_red[4], tmax_red[4];
    __shared__ __attribute__((aligned(16))) float epi_s[3 * 16];
    __shared__ __attribute__((aligned(16))) float epi0_s[3 * 8];
    ConvArgs a = a_in;
    FirstArgs f = f_in;
    // every argument the kernel uses in ONE batch of scalar loads (as conv3_split_kernel does: left alone, each is fetched where it is first needed)
    asm volatile("" : "+s"(a.X), "+s"(a.Y), "+s"(a.Z), "+s"(a.tilesX), "+s"(a.tilesY), "+s"(a.nxcd), "+s"(a.xper), "+s"(a.xrem), "+s"(a.mdiv[2]), "+s"(a.mdiv[3]),
                      "+s"(a.mdiv[4]), "+s"(a.nx0), "+s"(a.nx1), "+s"(a.ny0), "+s"(a.ny1), "+s"(a.sx0), "+s"(a.sx1), "+s"(a.sy0), "+s"(a.sy1), "+s"(a.pz), "+s"(a.PX),
                      "+s"(a.PY), "+s"(a.PZ), "+s"(a.act), "+s"(a.pg_yz), "+s"(a.pg_z), "+s"(a.mdivp[0]), "+s"(a.mdivp[1]), "+s"(a.p_first)
                    : "s"(a_in.out), "s"(a_in.pool), "s"(a_in.epi), "s"(a_in.amax_out), "s"(a_in.wpack), "s"(f_in.vol), "s"(f_in.wf16), "s"(f_in.epi));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t b = blockIdx.x;
    auto divmod = [](uint32_t& n, uint32_t d, uint32_t m) { uint32_t q = __umulhi(n, m), r = n - q * d; if (r >= d) { ++q; r -= d; }
                                                             n = q; return (int)r; };
    {
        const uint32_t xcd = (uint32_t)divmod(b, (uint32_t)a.nxcd, a.mdiv[4]);
        b += xcd * a.xper + (xcd < a.xrem ? xcd : a.xrem);
    }
    const int ty = divmod(b, (uint32_t)a.tilesY, a.mdiv[2]);
    const int tx = divmod(b, (uint32_t)a.tilesX, a.mdiv[3]);
    const int p = (int)b;
    const int x0 = tx * G::TXv, y0 = ty * G::TYv;
    const int g = lane >> 4, zl = lane & 15;
    const TileGeom& q = f.q;
    u32x4* const wlds = reinterpret_cast<u32x4*>(lds + G::PLANE);
    static_assert(320 * 16 <= G::PLANE, "L0 weights fit into the second plane");
    const u32x4 w_pre0 = f.wf16[tid];
    u32x4 w_pre1 = u32x4{0u, 0u, 0u, 0u};
    if (tid < 64) w_pre1 = f.wf16[256 + tid];
    {
        // the patch's place in the volume's patch grid: scalar, by the host's reciprocals (a.p_first = f.p_begin, a.pg_yz = gy * gz, a.pg_z = gz)
        uint32_t pgq = (uint32_t)(a.p_first + p);
        const int rem = divmod(pgq, (uint32_t)a.pg_yz, a.mdivp[0]);
        uint32_t jq = (uint32_t)rem;
        const int pk = divmod(jq, (uint32_t)a.pg_z, a.mdivp[1]);
        const int pi = (int)pgq, pj = (int)jq;
        if (tid < IX) { const int l = x0 - 2 + tid; mapx[tid] = (l >= 0 && l < q.nx) ? reflect_idx(pi * q.cx + l - q.bx, q.vx) : -1; }
        else if (tid >= 64 && tid < 64 + IY) { const int t = tid - 64, l = y0 - 2 + t; mapy[t] = (l >= 0 && l < q.ny) ? reflect_idx(pj * q.cy + l - q.by, q.vy) : -1; }
        // (z: BYTE offsets; 'outside the patch' = an offset beyond the volume -- the gather reads through a buffer descriptor, which returns 0 there)
        else if (tid >= 128 && tid < 128 + IZ) { const int t = tid - 128, l = t - 1; mapz[t] = (l >= 0 && l < q.nz) ? reflect_idx(pk * q.cz + l - q.bz, q.vz) * 4 : 0x40000000; }
        else if (tid >= 192 && tid < 192 + 48) { const int t = tid - 192; epi_s[t] = a.epi[(t >> 4) * (a.nt_total * 16) + (t & 15)]; }
        if (tid >= 32 && tid < 32 + 24) epi0_s[tid - 32] = f.epi[tid - 32];
    }
    __syncthreads();
    // ---- gather of the 1-channel input tile (thread t < 2 * IX * IY: half a z column), its maximum, packed (hi, lo) fp16 pairs
    constexpr int HALF = IZ / 2;
    static_assert(IZ % 2 == 0 && 2 * IX * IY <= 256, "gather mapping");
    float vals[HALF]; float tmax = 0.f;
    const int gcol = tid >> 1, gz0 = (tid & 1) * HALF;
    const bool gact = tid < 2 * IX * IY;
    {
        // one 32-bit offset per element on a buffer descriptor of the volume: row offset + z offset, either of them beyond the volume where the
        // voxel is L0's zero padding (0x80000000 + 0x40000000 does not wrap into range; the launch requires a volume below 1 GiB) -- the
        // hardware's range check supplies the zero: no select, no 64-bit address arithmetic
        const int hx = gcol / IY, hy = gcol - hx * IY;
        const int sx = gact ? mapx[hx] : -1, sy = gact ? mapy[hy] : -1;
        const uint32_t rowoff = (sx >= 0 && sy >= 0) ? (uint32_t)((sx * q.vy + sy) * q.vz) * 4u : 0x80000000u;
        const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f.vol), 0, q.vx * q.vy * q.vz * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const uint32_t off = rowoff + (uint32_t)mapz[gz0 + i];
            const float v = ((CT_ABL) & 8) ? (float)(off & 7) : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vrs, off, 0, 0));
            vals[i] = v; tmax = fmaxf(tmax, fabsf(v));
        }
    }
    tmax = wave_max_nonneg_l63(tmax);
    if (lane == 63) tmax_red[wave] = tmax;
    // L0's packed weights (5 K-blocks x 64 lanes x 16 B) were requested at kernel entry; parked in the second plane (free until L0 is done)
    // they are LDS reads in the K loop instead of five dependent L2 round trips per wave
    wlds[tid] = w_pre0;
    if (tid < 64) wlds[256 + tid] = w_pre1;
    __syncthreads();
    float out_mul0, out_mul, in_scale;
    {
        const float tmaxv = fmaxf(fmaxf(tmax_red[0], tmax_red[1]), fmaxf(tmax_red[2], tmax_red[3]));
        const int kexp = amax_exponent(__float_as_uint(tmaxv));
        const float in_scale0 = pow2f(-kexp);
        out_mul0 = pow2f(kexp) * f.wscale_inv;
        const int k1e = amax_exponent(__float_as_uint(__builtin_fmaf(f.bound_a, tmaxv, f.bound_b)));    // L1's input scale from the bound of L0's outputs
        in_scale = pow2f(-k1e);
        out_mul = pow2f(k1e) * a.wscale_inv;
        if (gact) {
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                const float x = vals[i] * in_scale0;
                const float hi = __uint_as_float(__float_as_uint(x) & 0xffffe000u);
                const uint32_t pk = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(hi, x - hi));   // [15:0] hi, [31:16] lo
                itile[gcol * IZ + gz0 + i] = uint2{pk, pk & 0xffffu};          // the B fragment's (a_hi, a_lo, a_hi, 0) as it is read
            }
        }
        if (tid == 255) itile[IX * IY * IZ] = uint2{0u, 0u};                  // what the pad slot of the tile's last column reads (weight 0: must be finite)
    }
    __syncthreads();
    // ---- L0 on the halo: pair column c = wave + 4 m (x pair c / 10, y c % 10; slots 30, 31 repeat pair 29), rows (x-select, cout), lane group g = dx'
    f32x4 acc0[8];
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(epi0_s + 4 * (g & 1)) * rcp_pow2(out_mul0);      // the bias as the accumulators' initial value
#pragma unroll
        for (int m = 0; m < 8; ++m) acc0[m] = b0;
    }
    int pxi_[8], hy_[8];                                                      // (scalar: wave is uniform)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int c = min(wave + 4 * m, NPAIR - 1);
        pxi_[m] = (c * 205) >> 11;                                            // c / 10 for c < 69
        hy_[m] = c - pxi_[m] * G::HYv;
    }
    if constexpr (!((CT_ABL) & 16384)) {
        const int lane_l0 = ((g * IY) * IZ + zl) * 8;
        int cbl[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) cbl[m] = lane_l0 + ((2 * pxi_[m]) * IY + hy_[m]) * (IZ * 8);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const u32x4 wj = wlds[j * 64 + lane];
            // (a compiler fence per K-block: left alone, the load-merging pass pairs tap reads of DIFFERENT K-blocks into one ds_read2_b64 and pays
            //  six v_mov per column to sort the halves out again)
            asm volatile("" ::: "memory");
            u32x4 av[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const char* ib = lds + cbl[m];
                const uint2 t0 = *reinterpret_cast<const uint2*>(ib + l0_tap_entry(2 * j, IZ) * 8);
                const uint2 t1 = *reinterpret_cast<const uint2*>(ib + l0_tap_entry(2 * j + 1, IZ) * 8);
                av[m] = u32x4{t0.x, t0.y, t1.x, t1.y};
            }
            // all eight fragment reads go out before the first MFMA (left alone, the scheduler issues each just in time and every MFMA eats an LDS round trip)
            asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 8; ++m)
                acc0[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wj), __builtin_bit_cast(f16x8, av[m]), acc0[m], 0, 0, 0);
        }
    }
    const float alpha = a.act == 0 ? kLeakyAlpha : 0.f;
    const float al2 = alpha;
    {
        EpiQuad k0 = epi_quad_load(epi0_s, 8, 4 * (g & 1));
        k0.s *= out_mul0 * in_scale; k0.h *= in_scale;                        // values leave the epilogue in L1's scaled units (powers of two: exact)
        float unused = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) acc0[m] = epi_quad_apply_c(acc0[m], al2, k0, unused);
        // outside the patch the tile holds L1's zero padding: only tiles on the patch border have such voxels (wave-uniform test)
        if (!(x0 >= 1 && x0 + G::HXv - 1 <= a.X && y0 >= 1 && y0 + G::HYv - 1 <= a.Y)) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int px = x0 - 1 + 2 * pxi_[m] + (g >> 1), py = y0 - 1 + hy_[m];
                const bool inside = px >= 0 && px < a.X && py >= 0 && py < a.Y;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0[m][e] = inside ? acc0[m][e] : 0.f;
            }
        }
    }
    __syncthreads();                                                          // every wave is done with the input tile and L0's weights: the planes are free
    {
        if (tid < 4 * G::HXv * G::HYv) {          // the z-halo rows of L1's tile are its 'same' padding (one z block): zeros in both planes
            const int col = tid >> 2, row = (tid & 2) ? HZg - 1 : 0, plane = tid & 1;
            *reinterpret_cast<uint4*>(lds + plane * G::PLANE + (col * HZg + row) * 16) = uint4{0u, 0u, 0u, 0u};
        }
        const int lane_st = ((((g >> 1) * HYg) * HZg + 1 + zl) * 2 + (g & 1)) * 8;
        float one = 1.f;
        asm volatile("" : "+v"(one));                                         // (opaque: see h_split4_scaled)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            uint2 hh, ll;
            h_split4_scaled(acc0[m], one, hh, ll);
            char* d = lds + lane_st + ((2 * pxi_[m]) * HYg + hy_[m]) * (HZg * 16);
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + G::PLANE) = ll;
        }
    }
    __syncthreads();
    // ---- L1: the ordinary single-chunk MFMA phase of conv3_split_kernel<true, 1, false, false, false>
    constexpr int NT = 1;
    const int wx = 2 * (wave >> 1), wy = 4 * (wave & 1);
    const int lanepos = (wx * HYg + wy) * HZg + zl;
    f32x4 acc[8][NT];
    {
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(epi_s + 4 * g) * rcp_pow2(out_mul);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt][0] = b1;
    }
    {
        int tapoff[KB_STD];
        bf_tap_offsets<KB_STD, false, false, false>(lanepos, g, tapoff);
        bf_chunk_mma<true, NT, 8, KB_STD, false, false, false, false>(acc, lds, tapoff, reinterpret_cast<const uint4*>(a.wpack), (uint32_t)lane * 16u, a.nt_total);
    }
    // ---- L1 epilogue.  Preconditions of the launch (run_network): X % 4 == 0, Y % 8 == 0, Z == 16, Cout == 16 -- every lane holds a valid output.
    const int z = zl;
    const int xw = x0 + wx, yw = y0 + wy;                                     // the wave's 2 x 4 block of columns
    float vmax = 0.f;
    {
        EpiQuad k1 = epi_quad_load(epi_s, 16, 4 * g);
        k1.s *= out_mul;
        if (xw >= a.nx0 && xw + 2 <= a.nx1 && yw >= a.ny0 && yw + 4 <= a.ny1) {           // (wave-uniform) every column enters the tensor's maximum
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt][0] = epi_quad_apply_c(acc[mt][0], al2, k1, vmax);
        } else {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                float cmax = 0.f;
                acc[mt][0] = epi_quad_apply_c(acc[mt][0], al2, k1, cmax);
                const int x = xw + (mt >> 2), y = yw + (mt & 3);
                if (x >= a.nx0 && x < a.nx1 && y >= a.ny0 && y < a.ny1) vmax = fmaxf(vmax, cmax);
            }
        }
    }
    const int OQ = a.cout >> 3;
    // stores through a buffer descriptor of this PATCH's output tensor: scalar column offset (SALU) + a 32-bit lane offset that is the same
    // for all eight columns -- no vector instruction per store
    const uint32_t lane_off = (uint32_t)((((g >> 1) * a.Z + z) * 8 + 4 * (g & 1)) * 4);
    const int col_bytes = OQ * a.Z * 32;
    if (a.out && !((CT_ABL) & 2048)) {
        const size_t patch_bytes = (size_t)a.X * a.Y * col_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + (size_t)p * patch_bytes, 0,
                                                                            (int)patch_bytes, 0x00020000);
        const int row0 = (xw * a.Y + yw) * col_bytes, row1 = row0 + a.Y * col_bytes;
        if (xw >= a.sx0 && xw + 2 <= a.sx1 && yw >= a.sy0 && yw + 4 <= a.sy1) {           // (wave-uniform) the whole block lies inside the stored window
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][0]), rs, lane_off, (mt >> 2 ? row1 : row0) + (mt & 3) * col_bytes, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int x = xw + (mt >> 2), y = yw + (mt & 3);
                if (x >= a.sx0 && x < a.sx1 && y >= a.sy0 && y < a.sy1)     // (wave-uniform)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mt][0]), rs, lane_off, (mt >> 2 ? row1 : row0) + (mt & 3) * col_bytes, 0);
            }
        }
    }
    // (the per-patch maximum -- wave reduction, barrier, one atomic -- after the output stores have been issued: they drain meanwhile)
    if (a.amax_out) amax_publish(vmax, a.amax_out + p * AMAX_STRIDE, tid, amax_red);
    if (a.pool && !((CT_ABL) & 2048)) {      // MaxPooling3D (2, 2, pz): the wave's 2 x 4 columns are two 2 x 2 blocks
        const int pcol_bytes = OQ * a.PZ * 32;
        const size_t ppatch_bytes = (size_t)a.PX * a.PY * pcol_bytes;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.pool) + (size_t)p * ppatch_bytes, 0,
                                                                             (int)ppatch_bytes, 0x00020000);
        const int prow = ((xw >> 1) * a.PY + (yw >> 1)) * pcol_bytes;
        f32x4 m[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                m[blk][e] = max4_nc(acc[2 * blk][0][e], acc[2 * blk + 1][0][e], acc[4 + 2 * blk][0][e], acc[5 + 2 * blk][0][e]);
        if (a.pz == 2) {                                                      // (uniform) z neighbours sit in adjacent lanes
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 4; ++e) m[blk][e] = max2_nc(m[blk][e], __shfl_xor(m[blk][e], 1));
            const uint32_t plane_off = (uint32_t)((((g >> 1) * a.PZ + (z >> 1)) * 8 + 4 * (g & 1)) * 4);
            if ((zl & 1) == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m[0]), prs, plane_off, prow, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m[1]), prs, plane_off, prow + pcol_bytes, 0);
            }
        } else {
            const uint32_t plane_off = (uint32_t)((((g >> 1) * a.PZ + z) * 8 + 4 * (g & 1)) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m[0]), prs, plane_off, prow, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m[1]), prs, plane_off, prow + pcol_bytes, 0);
        }
    }
}

// blocked [X][Y][C/8][Z][8] (patch 0) -> Keras NDHWC [X][Y][Z][C]   (parity tests only)
__global__ __launch_bounds__(256) void unblock_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      int X, int Y, int Z, int C) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)X * Y * Z * C;
    if (gid >= n) return;
    const int c = (int)(gid % C); size_t r = gid / C;
    const int z = (int)(r % Z); r /= Z;
    const int y = (int)(r % Y); const int x = (int)(r / Y);
    dst[gid] = src[((((size_t)x * Y + y) * (C >> 3) + (c >> 3)) * Z + z) * 8 + (c & 7)];
}

// ------------------------------------------------------------------------------------------------
// host side: architecture tables, weight packing, the layer program
// ------------------------------------------------------------------------------------------------
struct ArchDesc {
    int in[3]; int pool[3]; int act; int ndown; int down[3][2]; int up[3][2]; int outw[2];
};
const ArchDesc kArch[3] = {
    {{160, 160, 16}, {2, 2, 1}, 0, 3, {{8, 16}, {16, 32}, {32, 64}}, {{64, 64}, {32, 32}, {16, 16}}, {8, 8}},   // unet3d.py:26-37,84-98
    {{96, 96, 8},    {2, 2, 1}, 1, 2, {{64, 64}, {128, 128}, {0, 0}}, {{256, 256}, {128, 128}, {0, 0}}, {64, 64}}, // unet3d.py:40-67
    {{64, 64, 64},   {2, 2, 2}, 0, 3, {{8, 16}, {16, 32}, {32, 64}}, {{64, 64}, {32, 32}, {16, 16}}, {8, 8}},   // unet3d.py:70-81
};

struct ConvPlan {
    int cin, cout, NT;    // NT = cout tiles of 16 per block (<= 4)
    int nt_total;
    int level;            // resolution level of the conv
    int srcA, srcB;       // tensor ids (srcA = -1 when there is no concat)
    int CA, CB;
    int dst;              // tensor id or -1 (head layer)
    int pool_dst;         // tensor id or -1
    bool head;
    bool c8;              // Cout == 8: paired-column kernel
    bool fold;            // decoder conv over concat([upsample(low), skip]): folded taps for the upsampled channels
    bool bf;              // split (bf16x6 / f16x3) matrix-pipe kernel instead of the f32-input MFMA kernel
    bool f16;             // the split kernel's f16x3 variant (2 fp16 components, 3 products, per-patch power-of-two scaling)
    float wscale_inv;     // f16x3: 1 / power-of-two scale of the packed weights
    int nt_used;          // instantiation launched by the last run (small grids split NT = 4 into 2 x NT = 2)
    int tile[3];          // workgroup tile of the last run
    int region[4];        // x0, x1, y0, y1 computed by the last run (volume path: the part the centre crops depend on)
    int needed[4];        // the part of it some kept voxel really depends on (region = needed rounded out to whole tiles)
    int region_e[4], needed_e[4];   // the same for patches on the volume's far faces (their kept crop is shorter)
    int store[4];         // x0, x1, y0, y1 of the full-resolution output that has a reader (stores outside it are skipped)
    size_t wpack_off;     // float4 offset into the device weight arena
    size_t epi_off;       // float offset
};
struct TensorPlan { int level; int C; size_t off; /* floats per patch offset */ int amax_slot; /* pooled tensors share their parent's */ };

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

}  // namespace

struct ct_unet {
    int arch_id, device;
    ArchDesc ad;
    int nlevels;
    int dims[4][3];
    std::vector<ConvPlan> convs;     // convs[0] is the Cin=1 first conv
    std::vector<TensorPlan> tensors; // tensor 0 = input patches (C=1), last = prob out is external
    size_t floats_per_patch;         // workspace floats per patch (all intermediates)
    float* d_weights;                // device arena
    size_t first_w_off, head_off;    // float offsets
    size_t first_mfma_off;           // packed weights of conv_first_mfma_kernel (Cout == 8)
    size_t first_f16_off;            // packed (hi, hi, lo, 0) fp16 weights of conv_first_f16_kernel
    size_t first_f16b_off;           // the same weights in conv_l0l1_fused_kernel's K order
    float first_bound_a, first_bound_b;   // |first conv's output| <= a * max|input| + b (conv_l0l1_fused_kernel scales the second conv's input by it)
    float first_wscale_inv;          // 1 / their power-of-two scale
    bool first_f16;                  // the split-fp16 first conv is in use
    bool fused01 = false;            // the last run evaluated the first conv inside the second one's workgroups (conv_l0l1_fused_kernel)
    size_t arena_floats;
    // optional per-launch HIP-event timing (bench.py roofline): pairs recorded on the launch stream
    bool timing;
    std::vector<hipEvent_t> ev_pool;          // 2 events per timed launch
    std::vector<int> ev_layer;                // conv index of each timed launch
};

namespace {

void conv_layer_list(const ArchDesc& ad, std::vector<std::pair<int, int>>& layers) {
    int c = 1; std::vector<int> skips;
    for (int i = 0; i < ad.ndown; ++i) {
        layers.push_back({c, ad.down[i][0]}); layers.push_back({ad.down[i][0], ad.down[i][1]});
        skips.push_back(ad.down[i][1]); c = ad.down[i][1];
    }
    for (int i = 0; i < ad.ndown; ++i) {
        layers.push_back({c, ad.up[i][0]}); layers.push_back({ad.up[i][0], ad.up[i][1]});
        c = ad.up[i][1] + skips[ad.ndown - 1 - i];
    }
    layers.push_back({c, ad.outw[0]}); layers.push_back({ad.outw[0], ad.outw[1]});
}

// Pack Keras kernel (3,3,3,Cin,Cout) into MFMA operand order:
//   wpack[chunk][slab][nt][lane = g*16 + n][t] = K[tap = 2*slab + (g>>1)][cin = 8*chunk + 4*(g&1) + t][cout = 16*nt + n]
// for slabs 0..12; slab 13 is a half slab: [t < 2] = K[tap 26][cin = 8*chunk + 2*g + t][cout] (0 for cout >= Cout).
void pack_conv_weights(const float* k, int cin, int cout, int NT, float* dst) {
    const int nchunks = cin / 8;
    for (int ch = 0; ch < nchunks; ++ch)
        for (int s = 0; s < NSLAB; ++s)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 4; ++t) {
                        const int g = lane >> 4, n = lane & 15;
                        int tap = 2 * s + (g >> 1), ci = 8 * ch + 4 * (g & 1) + t;
                        const int co = 16 * nt + n;
                        if (s == NSLAB - 1) { tap = (t < 2) ? 26 : 27; ci = 8 * ch + 2 * g + t; }      // half slab: cin 2g + t of tap 26
                        float v = 0.f;
                        if (tap < 27 && co < cout) v = k[((size_t)tap * cin + ci) * cout + co];
                        dst[((((size_t)ch * NSLAB + s) * NT + nt) * 64 + lane) * 4 + t] = v;
                    }
}

// Cout = 8 packing (conv3_mfma_c8_kernel): rows n = xs * 8 + co, taps dx' in 0..3:
//   wpack8[chunk][slab][lane = g*16 + n][t] = K[dx' - xs][dy][dz][8*chunk + 4*(g&1) + t][co], tap' = 2*slab + (g>>1)
void pack_conv_weights_c8(const float* k, int cin, float* dst) {
    const int nchunks = cin / 8;
    for (int ch = 0; ch < nchunks; ++ch)
        for (int s = 0; s < NSLAB8; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 4; ++t) {
                    const int g = lane >> 4, n = lane & 15;
                    const int tapp = 2 * s + (g >> 1), ci = 8 * ch + 4 * (g & 1) + t;
                    const int xs = n >> 3, co = n & 7;
                    const int dxp = tapp / 9, dy = (tapp / 3) % 3, dz = tapp % 3;
                    const int dx = dxp - xs;
                    float v = 0.f;
                    if (dx >= 0 && dx <= 2) v = k[((size_t)((dx * 3 + dy) * 3 + dz) * cin + ci) * 8 + co];
                    dst[(((size_t)ch * NSLAB8 + s) * 64 + lane) * 4 + t] = v;
                }
}

// Which original taps k (0..2) coincide on folded tap i (0..1) for output parity par: even -> {0 | 1,2}, odd -> {0,1 | 2}.
inline bool fold_member(int par, int i, int k) { return par == 0 ? (i == 0 ? k == 0 : k >= 1) : (i == 0 ? k <= 1 : k == 2); }

// conv3_mfma_fold_kernel packing.  Upsampled chunks (cin < CA), per parity class cls = px*2 + py:
//   wfold[chunk][cls][slab][nt][lane = g*16 + n][t] = sum over coinciding (kx, ky) of K[kx][ky][dz][cin][cout],
//   folded tap j = 2*slab + (g>>1) = (dxi*2 + dyi)*3 + dz, cin = 8*chunk + 4*(g&1) + t, cout = 16*nt + n;
// skip chunks follow in pack_conv_weights order.
void pack_conv_weights_fold(const float* k, int cin, int cout, int NT, int CA, float* dst) {
    const int nA = CA / 8;
    for (int ch = 0; ch < nA; ++ch)
        for (int cls = 0; cls < 4; ++cls)
            for (int s = 0; s < NFSLAB; ++s)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int t = 0; t < 4; ++t) {
                            const int g = lane >> 4, n = lane & 15, px = cls >> 1, py = cls & 1;
                            const int j = 2 * s + (g >> 1), dxi = j / 6, dyi = (j / 3) % 2, dz = j % 3;
                            const int ci = 8 * ch + 4 * (g & 1) + t, co = 16 * nt + n;
                            double v = 0.0;
                            if (co < cout)
                                for (int kx = 0; kx < 3; ++kx)
                                    for (int ky = 0; ky < 3; ++ky)
                                        if (fold_member(px, dxi, kx) && fold_member(py, dyi, ky))
                                            v += (double)k[((size_t)((kx * 3 + ky) * 3 + dz) * cin + ci) * cout + co];
                            dst[(((((size_t)ch * 4 + cls) * NFSLAB + s) * NT + nt) * 64 + lane) * 4 + t] = (float)v;
                        }
    float* d2 = dst + (size_t)nA * 4 * NFSLAB * NT * 64 * 4;
    const int nB = (cin - CA) / 8;
    for (int ch = 0; ch < nB; ++ch)
        for (int s = 0; s < NSLAB; ++s)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 4; ++t) {
                        const int g = lane >> 4, n = lane & 15;
                        int tap = 2 * s + (g >> 1), ci = CA + 8 * ch + 4 * (g & 1) + t;
                        const int co = 16 * nt + n;
                        if (s == NSLAB - 1) { tap = (t < 2) ? 26 : 27; ci = CA + 8 * ch + 2 * g + t; }
                        float v = 0.f;
                        if (tap < 27 && co < cout) v = k[((size_t)tap * cin + ci) * cout + co];
                        d2[((((size_t)ch * NSLAB + s) * NT + nt) * 64 + lane) * 4 + t] = v;
                    }
}
inline size_t fold_pack_floats(int cin, int NT, int CA) {
    return ((size_t)(CA / 8) * 4 * NFSLAB + (size_t)((cin - CA) / 8) * NSLAB) * NT * 64 * 4;
}

// conv3_mfma_c8_fold_kernel packing.  Upsampled chunks, per y parity py: rows n = xs*8 + co,
//   wfold8[chunk][py][slab][lane][t]: folded tap j = 2*slab + (g>>1) = (dxl*2 + dyi)*3 + dz, dxl = low-res x offset + 1;
//   row xs = 0 (even x): dxl 0 <- kx {0}, dxl 1 <- kx {1, 2};  row xs = 1 (odd x): dxl 1 <- kx {0, 1}, dxl 2 <- kx {2}.
void pack_conv_weights_c8_fold(const float* k, int cin, int CA, float* dst) {
    const int nA = CA / 8;
    for (int ch = 0; ch < nA; ++ch)
        for (int py = 0; py < 2; ++py)
            for (int s = 0; s < NFSLAB8; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 4; ++t) {
                        const int g = lane >> 4, n = lane & 15, xs = n >> 3, co = n & 7;
                        const int j = 2 * s + (g >> 1), dxl = j / 6, dyi = (j / 3) % 2, dz = j % 3;
                        const int ci = 8 * ch + 4 * (g & 1) + t;
                        const int dxi = dxl - xs;                     // folded x tap index of this row's parity (0 or 1), else none
                        double v = 0.0;
                        if (dxi >= 0 && dxi <= 1)
                            for (int kx = 0; kx < 3; ++kx)
                                for (int ky = 0; ky < 3; ++ky)
                                    if (fold_member(xs, dxi, kx) && fold_member(py, dyi, ky))
                                        v += (double)k[((size_t)((kx * 3 + ky) * 3 + dz) * cin + ci) * 8 + co];
                        dst[((((size_t)ch * 2 + py) * NFSLAB8 + s) * 64 + lane) * 4 + t] = (float)v;
                    }
    float* d2 = dst + (size_t)nA * 2 * NFSLAB8 * 64 * 4;
    const int nB = (cin - CA) / 8;
    for (int ch = 0; ch < nB; ++ch)
        for (int s = 0; s < NSLAB8; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 4; ++t) {
                    const int g = lane >> 4, n = lane & 15;
                    const int tapp = 2 * s + (g >> 1), ci = CA + 8 * ch + 4 * (g & 1) + t;
                    const int xs = n >> 3, co = n & 7;
                    const int dxp = tapp / 9, dy = (tapp / 3) % 3, dz = tapp % 3;
                    const int dx = dxp - xs;
                    float v = 0.f;
                    if (dx >= 0 && dx <= 2) v = k[((size_t)((dx * 3 + dy) * 3 + dz) * cin + ci) * 8 + co];
                    d2[(((size_t)ch * NSLAB8 + s) * 64 + lane) * 4 + t] = v;
                }
}
inline size_t fold_pack_floats_c8(int cin, int CA) {
    return ((size_t)(CA / 8) * 2 * NFSLAB8 + (size_t)((cin - CA) / 8) * NSLAB8) * 64 * 4;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- bf16x6 packing ------------------------------------------------------------------------------------------
// effective fp32 weight of tap slot t / input channel ci / MFMA row `row` (cout index; Cout = 8: xs * 8 + co) for the
// four tap sets of conv3_bf16x6_kernel (see bf_tap_pos); cls = parity class of a folded chunk
float bf_weight(const float* k, int cin, int cout, bool c8, bool folded, int cls, int t, int ci, int row) {
    if (!c8 && row >= cout) return 0.f;
    const int xs = row >> 3, co8 = row & 7;
    if (!folded) {
        if (!c8) return t < 27 ? k[((size_t)t * cin + ci) * cout + row] : 0.f;
        if (t >= 36) return 0.f;
        const int dx = t / 9 - xs, dy = (t / 3) % 3, dz = t % 3;
        return (dx >= 0 && dx <= 2) ? k[((size_t)((dx * 3 + dy) * 3 + dz) * cin + ci) * 8 + co8] : 0.f;
    }
    int px, py, dxi, dyi, dz, co, cstride;
    if (!c8) {
        if (t >= 12) return 0.f;
        px = cls >> 1; py = cls & 1; dxi = t / 6; dyi = (t / 3) % 2; dz = t % 3; co = row; cstride = cout;
    } else {
        if (t >= 18) return 0.f;
        px = xs; py = cls; dxi = t / 6 - xs; dyi = (t / 3) % 2; dz = t % 3; co = co8; cstride = 8;
        if (dxi < 0 || dxi > 1) return 0.f;
    }
    double v = 0.0;
    for (int kx = 0; kx < 3; ++kx)
        for (int ky = 0; ky < 3; ++ky)
            if (fold_member(px, dxi, kx) && fold_member(py, dyi, ky))
                v += (double)k[((size_t)((kx * 3 + ky) * 3 + dz) * cin + ci) * cstride + co];
    return (float)v;
}

inline void bf_split_host(float x, uint16_t out[3]) {       // exact: x = h + m + l (truncation split, see bf_split4)
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t hb = u & 0xffff0000u; float h; memcpy(&h, &hb, 4);
    const float r = x - h; uint32_t ru; memcpy(&ru, &r, 4);
    const uint32_t mb = ru & 0xffff0000u; float m; memcpy(&m, &mb, 4);
    const float r2 = r - m; uint32_t lu; memcpy(&lu, &r2, 4);
    out[0] = (uint16_t)(hb >> 16); out[1] = (uint16_t)(mb >> 16); out[2] = (uint16_t)(lu >> 16);
}

inline void h_split_host(float x, uint16_t out[2]) {          // x ~ hi + lo in fp16 (round to nearest), |error| <= 2^-22 |x|
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    memcpy(&out[0], &hi, 2); memcpy(&out[1], &lo, 2);
}

// uint4 (16 B) units of one conv's packed weights (ncomp = 3 bf16 or 2 fp16 components)
inline size_t bf_pack_units(int cin, int nt_total, int CA, bool c8, bool fold, int ncomp) {
    const int nA = fold ? CA / 8 : 0, nB = cin / 8 - nA;
    const int kbs = c8 ? KB_C8 : KB_STD, kbf = c8 ? KB_C8F : KB_FOLD, ncls = c8 ? 2 : 4;
    return ((size_t)nA * ncls * kbf + (size_t)nB * kbs) * nt_total * ncomp * 64;
}

// wbf[section][kb][nt][comp][lane = g*16 + n][e] = component of W(tap slot 4 kb + g, cin 8 chunk + e, row 16 nt + n) * wscale
// (ncomp = 3: exact bf16 truncation split, wscale 1; ncomp = 2: fp16 hi/lo of the scaled weight)
void pack_conv_weights_bf(const float* k, int cin, int cout, int nt_total, int CA, bool c8, bool fold, uint16_t* dst,
                          int ncomp = 3, float wscale = 1.f) {
    const int nA = fold ? CA / 8 : 0, nchunks = cin / 8;
    const int kbs = c8 ? KB_C8 : KB_STD, kbf = c8 ? KB_C8F : KB_FOLD, ncls = c8 ? 2 : 4;
    size_t unit = 0;                                           // running uint4 index
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool folded = ch < nA;
        const int ncl = folded ? ncls : 1, KB = folded ? kbf : kbs;
        for (int cls = 0; cls < ncl; ++cls)
            for (int kb = 0; kb < KB; ++kb)
                for (int nt = 0; nt < nt_total; ++nt) {
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, n = lane & 15;
                        for (int e = 0; e < 8; ++e) {
                            uint16_t c3[3];
                            const float wv = bf_weight(k, cin, cout, c8, folded, cls, 4 * kb + g, 8 * ch + e, 16 * nt + n);
                            if (ncomp == 3) bf_split_host(wv, c3); else h_split_host(wv * wscale, c3);
                            for (int c = 0; c < ncomp; ++c) dst[((unit + (size_t)c * 64 + lane) * 8) + e] = c3[c];
                        }
                    }
                    unit += (size_t)ncomp * 64;
                }
    }
}

// largest |effective weight| of a conv (folded taps summed) -> power-of-two scale that puts it into [2^13, 2^14)
float f16_weight_scale(const float* k, int cin, int cout, int nt_total, int CA, bool c8, bool fold) {
    const int nA = fold ? CA / 8 : 0, nchunks = cin / 8;
    const int kbs = c8 ? KB_C8 : KB_STD, kbf = c8 ? KB_C8F : KB_FOLD, ncls = c8 ? 2 : 4;
    float m = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool folded = ch < nA;
        const int ncl = folded ? ncls : 1, KB = folded ? kbf : kbs;
        for (int cls = 0; cls < ncl; ++cls)
            for (int t = 0; t < 4 * KB; ++t)
                for (int e = 0; e < 8; ++e)
                    for (int row = 0; row < 16 * nt_total; ++row)
                        m = fmaxf(m, fabsf(bf_weight(k, cin, cout, c8, folded, cls, t, 8 * ch + e, row)));
    }
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int ex; frexpf(m, &ex);                                   // m = f * 2^ex, f in [0.5, 1)  ->  m * 2^(14 - ex) in [2^13, 2^14)
    int sh = 14 - ex; sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
    return ldexpf(1.f, sh);
}

template <int NT>
int launch_conv(const ConvArgs& a, int P, bool fold, hipStream_t st) {
    const int nblk = P * a.tilesX * a.tilesY * a.zblocks * a.ngroups;
    if (fold) hipLaunchKernelGGL((conv3_mfma_fold_kernel<NT>), dim3(nblk), dim3(256), 0, st, a);
    else      hipLaunchKernelGGL((conv3_mfma_kernel<NT>), dim3(nblk), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

template <bool F16, int NT>
int launch_conv_bf(const ConvArgs& a, int P, bool fold, bool z8, bool y10, hipStream_t st) {
    const int nblk = P * a.tilesX * a.tilesY * a.zblocks * a.ngroups;
    if (y10) {                                                // (the caller offers it for plain split-fp16 layers with NT <= 2 only)
        if constexpr (F16 && NT <= 2) hipLaunchKernelGGL((conv3_split_kernel<F16, NT, false, false, false, true>), dim3(nblk), dim3(256), 0, st, a);
        else return CT_ESHAPE;
    } else
    if (z8) {
        if (fold) hipLaunchKernelGGL((conv3_split_kernel<F16, NT, false, true, true>), dim3(nblk), dim3(256), 0, st, a);
        else      hipLaunchKernelGGL((conv3_split_kernel<F16, NT, false, false, true>), dim3(nblk), dim3(256), 0, st, a);
    } else {
        if (fold) hipLaunchKernelGGL((conv3_split_kernel<F16, NT, false, true, false>), dim3(nblk), dim3(256), 0, st, a);
        else      hipLaunchKernelGGL((conv3_split_kernel<F16, NT, false, false, false>), dim3(nblk), dim3(256), 0, st, a);
    }
    return (int)hipGetLastError();
}
template <bool F16>
int launch_conv_split(const ConvArgs& a_in, int P, int NTsel, bool c8, bool fold, bool z8, bool y10, hipStream_t st) {
    ConvArgs a = a_in;
    {
        const uint32_t nblk = (uint32_t)P * a.tilesX * a.tilesY * a.zblocks * (c8 ? 1 : a.ngroups);
        if (c8) a.ngroups = 1;
        if (a.nxcd < 1) a.nxcd = 1;
        a.xper = nblk / (uint32_t)a.nxcd; a.xrem = nblk - a.xper * (uint32_t)a.nxcd;
        const int d[5] = {a.ngroups, a.zblocks, a.tilesY, a.tilesX, a.nxcd};
        for (int i = 0; i < 5; ++i) a.mdiv[i] = d[i] <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)d[i]);
    }
    if (c8) {
        const int nblk = P * a.tilesX * a.tilesY * a.zblocks;
        if (fold) hipLaunchKernelGGL((conv3_split_kernel<F16, 1, true, true, false>), dim3(nblk), dim3(256), 0, st, a);
        else      hipLaunchKernelGGL((conv3_split_kernel<F16, 1, true, false, false>), dim3(nblk), dim3(256), 0, st, a);
        return (int)hipGetLastError();
    }
    switch (NTsel) {
        case 1: return launch_conv_bf<F16, 1>(a, P, fold, z8, y10, st);
        case 2: return launch_conv_bf<F16, 2>(a, P, fold, z8, y10, st);
        case 4: return launch_conv_bf<F16, 4>(a, P, fold, z8, y10, st);
        default: return CT_ESHAPE;
    }
}

struct TimedScope {     // records an event pair around one launch when the handle has timing enabled
    ct_unet* h; hipStream_t st; hipEvent_t e1 = nullptr;
    TimedScope(ct_unet* h_, int layer, hipStream_t st_);
    ~TimedScope() { if (e1) (void)hipEventRecord(e1, st); }
};

}  // namespace

TimedScope::TimedScope(ct_unet* h_, int layer, hipStream_t st_) : h(h_), st(st_) {
    if (!h->timing || h->ev_layer.size() >= 200000) return;
    hipEvent_t e0;
    if (hipEventCreate(&e0) != hipSuccess) return;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); e1 = nullptr; return; }
    h->ev_pool.push_back(e0); h->ev_pool.push_back(e1); h->ev_layer.push_back(layer);
    (void)hipEventRecord(e0, st);
}

extern "C" {

int ct_unet_num_conv_layers(const ct_unet_t* h) { return h ? (int)h->convs.size() : 0; }

int ct_unet_layer_info(const ct_unet_t* h, int layer, int* cin, int* cout, int dims_xyz[3], int* nt) {
    if (!h || layer < 0 || layer >= (int)h->convs.size()) return CT_EINVAL;
    const ConvPlan& c = h->convs[layer];
    if (cin) *cin = c.cin;
    if (cout) *cout = c.cout;
    // 0: first conv, -8: conv3_mfma_c8_kernel, -9: conv3_mfma_c8_fold_kernel, 1..4: conv3_mfma_kernel<nt>, 101..104: conv3_mfma_fold_kernel<nt - 100>
    // split kernels (conv3_split_kernel<f16, nt, c8, fold, z8>): the same codes +1000 for bf16x6 (Cout = 8: -1008 / -1009),
    // +2000 for f16x3 (-2008 / -2009)
    if (nt) {
        *nt = layer == 0 ? 0 : (c.c8 ? (c.fold ? -9 : -8) : (c.nt_used ? c.nt_used : c.NT) + (c.fold ? 100 : 0));
        if (layer > 0 && c.bf) *nt += (*nt < 0 ? -1 : 1) * (c.f16 ? 2000 : 1000);
    }
    if (dims_xyz) for (int i = 0; i < 3; ++i) dims_xyz[i] = h->dims[c.level][i];
    return CT_OK;
}

int ct_unet_layer_region(const ct_unet_t* h, int layer, int region[4]) {
    if (!h || !region || layer < 0 || layer >= (int)h->convs.size()) return CT_EINVAL;
    for (int i = 0; i < 4; ++i) region[i] = h->convs[layer].region[i];
    return CT_OK;
}

int ct_unet_layer_tile(const ct_unet_t* h, int layer, int tile_xyz[3]) {
    if (!h || !tile_xyz || layer < 0 || layer >= (int)h->convs.size()) return CT_EINVAL;
    for (int i = 0; i < 3; ++i) tile_xyz[i] = h->convs[layer].tile[i];
    return CT_OK;
}

int ct_unet_layer_fold_channels(const ct_unet_t* h, int layer) {
    if (!h || layer < 0 || layer >= (int)h->convs.size()) return CT_EINVAL;
    return h->convs[layer].fold ? h->convs[layer].CA : 0;
}

int ct_unet_set_timing(ct_unet_t* h, int enable) {
    if (!h) return CT_EINVAL;
    h->timing = enable != 0;
    return CT_OK;
}

// Synchronises the device, sums the elapsed time of every timed launch per conv layer, clears the log.
int ct_unet_get_timing(ct_unet_t* h, float* ms_per_layer, int* launches_per_layer, int n_layers) {
    if (!h || !ms_per_layer || !launches_per_layer || n_layers < (int)h->convs.size()) return CT_EINVAL;
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < n_layers; ++i) { ms_per_layer[i] = 0.f; launches_per_layer[i] = 0; }
    for (size_t k = 0; k < h->ev_layer.size(); ++k) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[2 * k], h->ev_pool[2 * k + 1]) == hipSuccess) {
            ms_per_layer[h->ev_layer[k]] += ms; launches_per_layer[h->ev_layer[k]] += 1;
        }
        (void)hipEventDestroy(h->ev_pool[2 * k]); (void)hipEventDestroy(h->ev_pool[2 * k + 1]);
    }
    h->ev_pool.clear(); h->ev_layer.clear();
    return CT_OK;
}

size_t ct_unet_num_weights(int arch_id) {
    if (arch_id < 0 || arch_id > 2) return 0;
    std::vector<std::pair<int, int>> layers;
    conv_layer_list(kArch[arch_id], layers);
    size_t n = 0;
    for (auto& l : layers) n += (size_t)27 * l.first * l.second + 5 * (size_t)l.second;
    return n + kArch[arch_id].outw[1] + 1;
}

int ct_unet_patch_shape(int arch_id, int s[3]) {
    if (arch_id < 0 || arch_id > 2 || !s) return CT_EINVAL;
    for (int i = 0; i < 3; ++i) s[i] = kArch[arch_id].in[i];
    return CT_OK;
}

size_t ct_unet_layer_dump_floats(int arch_id) {
    if (arch_id < 0 || arch_id > 2) return 0;
    const ArchDesc& ad = kArch[arch_id];
    std::vector<std::pair<int, int>> layers;
    conv_layer_list(ad, layers);
    size_t vox[4]; int d[3] = {ad.in[0], ad.in[1], ad.in[2]};
    for (int l = 0; l <= ad.ndown; ++l) { vox[l] = (size_t)d[0] * d[1] * d[2]; for (int i = 0; i < 3; ++i) d[i] /= ad.pool[i]; }
    size_t n = 0; int li = 0;
    for (int l = 0; l < ad.ndown; ++l) { n += vox[l] * layers[li++].second; n += vox[l] * layers[li++].second; }
    for (int k = 0; k < ad.ndown; ++k) { int l = ad.ndown - k; n += vox[l] * layers[li++].second; n += vox[l] * layers[li++].second; }
    n += vox[0] * layers[li++].second; n += vox[0] * layers[li++].second;
    return n;
}

int ct_unet_create(int arch_id, const float* w, size_t n_floats, int device, ct_unet_t** out) {
    if (arch_id < 0 || arch_id > 2 || !w || !out) return CT_EINVAL;
    if (n_floats != ct_unet_num_weights(arch_id)) return CT_ESHAPE;
    HIPCHK(hipSetDevice(device));
    ct_unet* h = new (std::nothrow) ct_unet();
    if (!h) return CT_EINVAL;
    h->timing = false;
    h->first_f16 = false; h->first_f16_off = 0; h->first_f16b_off = 0; h->first_wscale_inv = 1.f; h->first_bound_a = h->first_bound_b = 0.f;
    h->arch_id = arch_id; h->device = device; h->ad = kArch[arch_id];
    const ArchDesc& ad = h->ad;
    h->nlevels = ad.ndown + 1;
    for (int l = 0; l < h->nlevels; ++l)
        for (int i = 0; i < 3; ++i) {
            int v = ad.in[i]; for (int k = 0; k < l; ++k) v /= ad.pool[i];
            h->dims[l][i] = v;
        }
    std::vector<std::pair<int, int>> layers;
    conv_layer_list(ad, layers);

    // ---- tensor plan (bump allocation per patch; tensor 0 = input patches)
    auto vox = [&](int l) { return (size_t)h->dims[l][0] * h->dims[l][1] * h->dims[l][2]; };
    size_t off = 0;
    auto add_tensor = [&](int level, int C) {
        h->tensors.push_back({level, C, off, (int)h->tensors.size()});
        off += vox(level) * C;
        return (int)h->tensors.size() - 1;
    };
    int cur = add_tensor(0, 1);
    std::vector<int> skips;
    int li = 0;
    auto add_conv = [&](int level, int srcA, int CA, int srcB, int CB, int cout, bool pool, bool head) {
        ConvPlan c{};
        c.cin = CA + CB; c.cout = cout; c.nt_total = cout <= 16 ? 1 : cout / 16;
        c.NT = c.nt_total > 4 ? 4 : c.nt_total; c.level = level;
        c.srcA = srcA; c.srcB = srcB; c.CA = CA; c.CB = CB; c.head = head;
        c.dst = add_tensor(level, cout);     // the head layer's tensor is only written for parity dumps
        c.pool_dst = pool ? add_tensor(level + 1, cout) : -1;
        if (pool) h->tensors[c.pool_dst].amax_slot = h->tensors[c.dst].amax_slot;      // max-pooling cannot raise the maximum
        h->convs.push_back(c);
        ++li;
        return c.dst;
    };
    for (int l = 0; l < ad.ndown; ++l) {
        int t1 = add_conv(l, -1, 0, cur, h->tensors[cur].C, ad.down[l][0], false, false);
        int t2 = add_conv(l, -1, 0, t1, ad.down[l][0], ad.down[l][1], true, false);
        skips.push_back(t2);
        cur = h->convs.back().pool_dst;
    }
    int lowA = -1, lowCA = 0;       // pending low-res tensor to be upsampled+concatenated by the next conv
    for (int k = 0; k < ad.ndown; ++k) {
        const int l = ad.ndown - k;
        int t1 = (lowA < 0) ? add_conv(l, -1, 0, cur, h->tensors[cur].C, ad.up[k][0], false, false)
                            : add_conv(l, lowA, lowCA, cur, h->tensors[cur].C, ad.up[k][0], false, false);
        int t2 = add_conv(l, -1, 0, t1, ad.up[k][0], ad.up[k][1], false, false);
        lowA = t2; lowCA = ad.up[k][1];
        cur = skips[ad.ndown - 1 - k];          // concat([up(t2), skip])
    }
    int t1 = add_conv(0, lowA, lowCA, cur, h->tensors[cur].C, ad.outw[0], false, false);
    add_conv(0, -1, 0, t1, ad.outw[0], ad.outw[1], false, true);
    h->floats_per_patch = off;
    // the epilogues address a patch's tensor through a buffer descriptor with 32-bit offsets (CT_EPI_SBASE)
    for (const TensorPlan& tp : h->tensors) {
        const int* dd = h->dims[tp.level];
        if ((size_t)dd[0] * dd[1] * dd[2] * (size_t)tp.C * sizeof(float) >= (size_t)1 << 31) { delete h; return CT_ESHAPE; }
    }

    // ---- device weight arena: first conv [27][C0] + epi, then packed convs, then head
    std::vector<float> arena;
    auto push_epi = [&](const float* bias, const float* gamma, const float* beta, const float* mean, const float* var,
                        int cout, int CP) {
        size_t o = arena.size();
        arena.resize(o + 3 * (size_t)CP, 0.f);
        for (int c = 0; c < cout; ++c) {
            const float sc = gamma[c] / sqrtf(var[c] + kBnEps);
            arena[o + c] = bias[c];
            arena[o + CP + c] = sc;
            arena[o + 2 * CP + c] = beta[c] - mean[c] * sc;
        }
        return o;
    };
    const float* p = w;
    for (size_t i = 0; i < h->convs.size(); ++i) {
        ConvPlan& c = h->convs[i];
        const float* kern = p; p += (size_t)27 * c.cin * c.cout;
        const float* bias = p; p += c.cout;
        const float* gamma = p; p += c.cout;
        const float* beta = p; p += c.cout;
        const float* mean = p; p += c.cout;
        const float* var = p; p += c.cout;
        if (i == 0) {
            h->first_w_off = arena.size();
            arena.insert(arena.end(), kern, kern + (size_t)27 * c.cout);
            arena.resize(align_up(arena.size(), 4), 0.f);
            if (c.cout == 8) {      // MFMA packing: wfirst[t][lane = g*16 + n] = K[dx' - xs][dy][dz][co], tap' = 4t + g, n = xs*8 + co
                h->first_mfma_off = arena.size();
                arena.resize(arena.size() + 9 * 64, 0.f);
                for (int t = 0; t < 9; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, n = lane & 15, xs = n >> 3, co = n & 7;
                        const int kk = 4 * t + g, dxp = kk / 9, dy = (kk / 3) % 3, dz = kk % 3, dx = dxp - xs;
                        if (dx >= 0 && dx <= 2) arena[h->first_mfma_off + t * 64 + lane] = kern[((dx * 3 + dy) * 3 + dz) * 8 + co];
                    }
                // split-fp16 packing: wf16[j][lane = g*16 + n][4 dwords] = [w_hi, w_hi | w_lo, 0] of tap' 8j + 2g + s (s = 0, 1), row n
                float wmax = 0.f;
                for (int e = 0; e < 27 * 8; ++e) wmax = fmaxf(wmax, fabsf(kern[e]));
                float wscale = 1.f;
                if (wmax > 0.f && std::isfinite(wmax)) { int ex; frexpf(wmax, &ex); int sh = 14 - ex; sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh); wscale = ldexpf(1.f, sh); }
                h->first_wscale_inv = 1.f / wscale;
                h->first_f16_off = arena.size();
                arena.resize(arena.size() + 5 * 64 * 4, 0.f);
                uint16_t* wf = reinterpret_cast<uint16_t*>(arena.data() + h->first_f16_off);
                for (int j = 0; j < 5; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, n = lane & 15, xs = n >> 3, co = n & 7;
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const int kk = 8 * j + 2 * g + s2, dxp = kk / 9, dy = (kk / 3) % 3, dz = kk % 3, dx = dxp - xs;
                            uint16_t hl[2] = {0, 0};
                            if (kk < 36 && dx >= 0 && dx <= 2) h_split_host(kern[((dx * 3 + dy) * 3 + dz) * 8 + co] * wscale, hl);
                            uint16_t* d = wf + ((size_t)(j * 64 + lane) * 4 + 2 * s2) * 2;      // two dwords per sub-slot
                            d[0] = hl[0]; d[1] = hl[0]; d[2] = hl[1]; d[3] = 0;
                        }
                    }
                // the fused first pair (conv_l0l1_fused_kernel) orders L0's K differently: lane group g = dx' (x offset inside the output pair), K-block j
                // holds the (dy, dz) combinations 2j, 2j + 1 of the nine (slot 9: zero weights) -- the lane part of a B-fragment address is then dx' alone
                h->first_f16b_off = arena.size();
                arena.resize(arena.size() + 5 * 64 * 4, 0.f);
                uint16_t* wfb = reinterpret_cast<uint16_t*>(arena.data() + h->first_f16b_off);
                for (int j = 0; j < 5; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, n = lane & 15, xs = n >> 3, co = n & 7;
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const int idx = 2 * j + s2, dy = idx / 3, dz = idx % 3, dx = g - xs;
                            uint16_t hl[2] = {0, 0};
                            if (idx < 9 && dx >= 0 && dx <= 2) h_split_host(kern[((dx * 3 + dy) * 3 + dz) * 8 + co] * wscale, hl);
                            uint16_t* d = wfb + ((size_t)(j * 64 + lane) * 4 + 2 * s2) * 2;     // two dwords per sub-slot
                            d[0] = hl[0]; d[1] = hl[0]; d[2] = hl[1]; d[3] = 0;
                        }
                    }
                static const bool first_f16_on = !(getenv("CT_FIRST_F16") && atoi(getenv("CT_FIRST_F16")) == 0);
                const char* mathenv = getenv("CT_CONV_MATH");
                h->first_f16 = first_f16_on && !(mathenv && (strcmp(mathenv, "f32") == 0 || strcmp(mathenv, "bf16x6") == 0));
            }
            {   // bound of this layer's outputs: |LeakyReLU / ReLU (t)| <= |t|, so |out_c| <= |scale_c| (sum |w_c| max|in| + |bias_c|) + |shift_c|
                float ba = 0.f, bb = 0.f;
                for (int co = 0; co < c.cout; ++co) {
                    const float sc = gamma[co] / sqrtf(var[co] + kBnEps), sh = beta[co] - mean[co] * sc;
                    float sw = 0.f;
                    for (int t = 0; t < 27; ++t) sw += fabsf(kern[t * c.cout + co]);
                    ba = fmaxf(ba, fabsf(sc) * sw); bb = fmaxf(bb, fabsf(sc) * fabsf(bias[co]) + fabsf(sh));
                }
                h->first_bound_a = ba * 1.001f; h->first_bound_b = bb * 1.001f;      // (a margin for the fp32 rounding of the sums)
            }
            c.epi_off = push_epi(bias, gamma, beta, mean, var, c.cout, c.cout);
            arena.resize(align_up(arena.size(), 4), 0.f);
        } else {
            c.wpack_off = arena.size();
            // decoder conv over concat([UpSampling3D(2, 2, *)(low), skip]): fold the coinciding taps (CT_CONV_FOLD=0: off)
            static const bool fold_on = !(getenv("CT_CONV_FOLD") && atoi(getenv("CT_CONV_FOLD")) == 0);
            c.fold = fold_on && c.srcA >= 0 && c.CA % 8 == 0 && ad.pool[0] == 2 && ad.pool[1] == 2 && c.pool_dst < 0
                     && h->dims[c.level][0] % 2 == 0 && h->dims[c.level][1] % 2 == 0;
            // CT_CONV_MATH (read when the model is created): f16x3 (default) | bf16x6 | f32
            const char* math = getenv("CT_CONV_MATH");
            c.bf = !(math && strcmp(math, "f32") == 0);
            c.f16 = c.bf && !(math && strcmp(math, "bf16x6") == 0);
            c.wscale_inv = 1.f;
            if (c.bf) {                                    // split kernels (all four tap sets)
                c.c8 = (c.cout == 8 && c.pool_dst < 0);
                const int ncomp = c.f16 ? 2 : 3;
                float wscale = 1.f;
                if (c.f16) { wscale = f16_weight_scale(kern, c.cin, c.cout, c.nt_total, c.CA, c.c8, c.fold); c.wscale_inv = 1.f / wscale; }
                const size_t units = bf_pack_units(c.cin, c.nt_total, c.CA, c.c8, c.fold, ncomp);
                arena.resize(arena.size() + units * 4);
                pack_conv_weights_bf(kern, c.cin, c.cout, c.nt_total, c.CA, c.c8, c.fold,
                                     reinterpret_cast<uint16_t*>(arena.data() + c.wpack_off), ncomp, wscale);
            } else if (c.cout == 8 && c.pool_dst < 0 && c.fold) {
                c.c8 = true;
                arena.resize(arena.size() + fold_pack_floats_c8(c.cin, c.CA));
                pack_conv_weights_c8_fold(kern, c.cin, c.CA, arena.data() + c.wpack_off);
            } else if (c.fold) {
                arena.resize(arena.size() + fold_pack_floats(c.cin, c.nt_total, c.CA));
                pack_conv_weights_fold(kern, c.cin, c.cout, c.nt_total, c.CA, arena.data() + c.wpack_off);
            } else if (c.cout == 8 && c.pool_dst < 0) {          // paired-column kernel
                c.c8 = true;
                arena.resize(arena.size() + (size_t)(c.cin / 8) * NSLAB8 * 64 * 4);
                pack_conv_weights_c8(kern, c.cin, arena.data() + c.wpack_off);
            } else {
                arena.resize(arena.size() + (size_t)(c.cin / 8) * NSLAB * c.nt_total * 64 * 4);
                pack_conv_weights(kern, c.cin, c.cout, c.nt_total, arena.data() + c.wpack_off);
            }
            c.epi_off = push_epi(bias, gamma, beta, mean, var, c.cout, c.nt_total * 16);
        }
    }
    {   // head: [CP] weights zero padded + bias
        const ConvPlan& last = h->convs.back();
        const int CP = last.nt_total * 16;
        h->head_off = arena.size();
        arena.resize(arena.size() + align_up(CP + 1, 4), 0.f);
        for (int c = 0; c < last.cout; ++c) arena[h->head_off + c] = p[c];
        arena[h->head_off + CP] = p[last.cout];
    }
    h->arena_floats = arena.size();
    hipError_t e = hipMalloc((void**)&h->d_weights, arena.size() * sizeof(float));
    if (e != hipSuccess) { delete h; return (int)e; }
    e = hipMemcpy(h->d_weights, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->d_weights); delete h; return (int)e; }
    *out = h;
    return CT_OK;
}

void ct_unet_destroy(ct_unet_t* h) {
    if (!h) return;
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    (void)hipFree(h->d_weights);
    delete h;
}

// per-patch |max| slots of every tensor (split-fp16 kernels), placed after the tensors of a batch: [n_tensors][P][AMAX_STRIDE] uint32
static size_t amax_words(const ct_unet_t* h, int P) { return h->tensors.size() * (size_t)P * AMAX_STRIDE; }

size_t ct_unet_workspace_bytes(const ct_unet_t* h, int n_patches) {
    if (!h || n_patches <= 0) return 0;
    // intermediates of every patch + one prob buffer [n][X][Y][Z] used by ct_unet_predict_volume
    const size_t vox0 = (size_t)h->dims[0][0] * h->dims[0][1] * h->dims[0][2];
    return ((h->floats_per_patch + vox0) * (size_t)n_patches + amax_words(h, n_patches)) * sizeof(float) + 256;
}

struct VolSource { const float* vol; TileGeom q; int p_begin; };

static int run_network(ct_unet_t* h, float* ws, int P, float* prob_out, float* layer_dump, hipStream_t st,
                       const VolSource* vsrc = nullptr, float* vol_out = nullptr) {
    // ws: tensor t of patch batch lives at ws + tensors[t].off * P  (each tensor is [P][...])
    auto tptr = [&](int t) { return ws + h->tensors[t].off * (size_t)P; };
    // per-patch |max| slots (split-fp16 kernels): after the tensors of this batch, zeroed per run
    uint32_t* amax = reinterpret_cast<uint32_t*>(ws + h->floats_per_patch * (size_t)P);
    auto aptr = [&](int t) { return amax + (size_t)h->tensors[t].amax_slot * P * AMAX_STRIDE; };
    bool any_f16 = false;
    for (const ConvPlan& c : h->convs) any_f16 = any_f16 || c.f16;
    if (any_f16) HIPCHK(hipMemsetAsync(amax, 0, amax_words(h, P) * sizeof(uint32_t), st));
    const ArchDesc& ad = h->ad;
    size_t dump_off = 0;
    // XCDs the stream may use: 32-CU words of its CU mask with at least one CU enabled (CT_CONV_XCD=0: no tile remapping)
    int nxcd = 8;
    {
        static const int xcd_env = getenv("CT_CONV_XCD") ? atoi(getenv("CT_CONV_XCD")) : -1;
        if (xcd_env >= 0) nxcd = xcd_env;
        else {
            uint32_t mask[8] = {0};
            if (hipExtStreamGetCUMask(st, 8, mask) == hipSuccess) {
                int n = 0;
                for (int w = 0; w < 8; ++w) n += mask[w] != 0;
                if (n > 0) nxcd = n;
            } else (void)hipGetLastError();
        }
    }
    // Volume path: unet3_prediction keeps only the centre crop of every patch (unet3d.py:246-254), so decoder outputs that no kept voxel
    // depends on are dead work.  Walk the layers backwards: a conv NEEDS what its consumers read of its output (their needed region
    // grown by the 3x3x3 footprint, halved through an upsampling, doubled through a pool; z is never cut) and COMPUTES the whole
    // tiles covering that.  Every needed output has all of its inputs needed -- hence computed -- upstream, so the kept voxels are what
    // the full evaluation gives.  The rest of a computed tile may be fed from voxels nobody computed: it is written but never read
    // by a needed output (an MFMA column is one voxel, so nothing mixes), and only needed outputs enter the per-patch maxima that
    // scale the next layer.  unet3_a, shrink (24,24,2): L13 49 %, L12 60 %, L11 64 %, L10 64 %, L9 80 % of their tiles, the rest in
    // full.  CT_CONV_CROP=1: inputs derived from the computed tiles instead (no computed output ever sees an uncomputed input;
    // L11 80 %, L10 90 %), CT_CONV_CROP=0: everything, as the reference does.
    static const bool z8_on = !(getenv("CT_CONV_Z8") && atoi(getenv("CT_CONV_Z8")) == 0);
    static const int crop_mode = getenv("CT_CONV_CROP") ? atoi(getenv("CT_CONV_CROP")) : 2;
    for (int edge = 0; edge < 2; ++edge) {                  // 0: ordinary patches, 1: patches on the far x / y faces of the volume (shorter kept crop)
        struct Reg { int lo[2], hi[2]; bool any; };
        const size_t nc = h->convs.size();
        std::vector<Reg> need(h->tensors.size(), Reg{{0, 0}, {0, 0}, false});
        auto join = [](Reg& r, const int lo[2], const int hi[2]) {
            for (int ax = 0; ax < 2; ++ax) {
                r.lo[ax] = r.any ? (lo[ax] < r.lo[ax] ? lo[ax] : r.lo[ax]) : lo[ax];
                r.hi[ax] = r.any ? (hi[ax] > r.hi[ax] ? hi[ax] : r.hi[ax]) : hi[ax];
            }
            r.any = true;
        };
        const bool cut = crop_mode > 0 && vsrc && !layer_dump;
        for (size_t ii = nc; ii-- > 0;) {
            ConvPlan& c = h->convs[ii];
            const int* d = h->dims[c.level];
            Reg n{{0, 0}, {0, 0}, false};
            if (cut && c.f16 && ii > 0) {
                if (c.head) {
                    const int keepx = edge ? vsrc->q.vx - (vsrc->q.gx - 1) * vsrc->q.cx : vsrc->q.cx, keepy = edge ? vsrc->q.vy - (vsrc->q.gy - 1) * vsrc->q.cy : vsrc->q.cy;
                    const int lo[2] = {vsrc->q.bx, vsrc->q.by}, hi[2] = {vsrc->q.bx + keepx, vsrc->q.by + keepy};
                    join(n, lo, hi);
                }
                if (c.dst >= 0 && !c.head && need[c.dst].any) join(n, need[c.dst].lo, need[c.dst].hi);
                if (c.pool_dst >= 0 && need[c.pool_dst].any) {
                    const int lo[2] = {need[c.pool_dst].lo[0] * ad.pool[0], need[c.pool_dst].lo[1] * ad.pool[1]};
                    const int hi[2] = {need[c.pool_dst].hi[0] * ad.pool[0], need[c.pool_dst].hi[1] * ad.pool[1]};
                    join(n, lo, hi);
                }
            }
            if (edge == 0) {                                   // what of the full-resolution output is ever read (consumers come later in the list)
                static const bool store_cut = !(getenv("CT_CONV_STORE_CUT") && atoi(getenv("CT_CONV_STORE_CUT")) == 0);
                const bool w = cut && store_cut && c.f16 && ii > 0 && c.dst >= 0 && !c.head && need[c.dst].any;
                for (int ax = 0; ax < 2; ++ax) {
                    c.store[2 * ax] = w ? (need[c.dst].lo[ax] < 0 ? 0 : need[c.dst].lo[ax]) : 0;
                    c.store[2 * ax + 1] = w ? (need[c.dst].hi[ax] > d[ax] ? d[ax] : need[c.dst].hi[ax]) : d[ax];
                }
            }
            const bool z8 = z8_on && c.bf && !c.c8 && d[2] <= 8;
            const int tile[2] = {z8 ? 8 : TX, TY};
            for (int ax = 0; ax < 2; ++ax) {
                int lo = n.any ? n.lo[ax] : 0, hi = n.any ? n.hi[ax] : d[ax];
                lo = lo < 0 ? 0 : lo; hi = hi > d[ax] ? d[ax] : hi;
                int* needed = edge ? c.needed_e : c.needed; int* region = edge ? c.region_e : c.region;
                needed[2 * ax] = lo; needed[2 * ax + 1] = hi;
                lo &= ~1;                                              // the tile grid starts at an even voxel (2 x 2 pools, parity classes of
                hi = lo + (hi - lo + tile[ax] - 1) / tile[ax] * tile[ax];   // the folded taps, x pairs of the Cout = 8 kernels), not at a tile multiple
                hi = hi > d[ax] ? d[ax] : hi;
                region[2 * ax] = lo; region[2 * ax + 1] = hi;
            }
            int* needed = edge ? c.needed_e : c.needed; const int* region = edge ? c.region_e : c.region;
            if (crop_mode == 1) for (int k = 0; k < 4; ++k) needed[k] = region[k];   // conservative: whatever is computed counts as needed
            if (ii == 0) continue;
            // What the kernel READS for its needed outputs.  Ordinary kernels: the 3 x 3 footprint.  The Cout = 8 kernels compute two
            // x-adjacent voxels per MFMA column over their common 4 x 3 footprint: each voxel also multiplies the neighbour's extra column
            // by a zero weight -- and 0 x (a NaN from a voxel nobody computed) is NaN, so that column has to hold computed values too.
            int nlo[2] = {needed[0], needed[2]}, nhi[2] = {needed[1], needed[3]};
            if (c.c8) { nlo[0] &= ~1; nhi[0] = (nhi[0] + 1) & ~1; }
            int ilo[2], ihi[2];
            for (int ax = 0; ax < 2; ++ax) {
                ilo[ax] = nlo[ax] - 1 < 0 ? 0 : nlo[ax] - 1;
                ihi[ax] = nhi[ax] + 1 > d[ax] ? d[ax] : nhi[ax] + 1;
            }
            join(need[c.srcB], ilo, ihi);
            if (c.srcA >= 0) {
                const int* da = h->dims[c.level + 1];
                const int u[2] = {ad.pool[0] == 2 ? 1 : 0, ad.pool[1] == 2 ? 1 : 0};
                int lo[2] = {ilo[0] >> u[0], ilo[1] >> u[1]}, hi[2] = {(ihi[0] + u[0]) >> u[0], (ihi[1] + u[1]) >> u[1]};
                if (c.c8 && u[0]) {                                // folded pairs: low-res offsets {-1, 0, +1} around the pair's own low-res voxel
                    lo[0] = (nlo[0] >> 1) - 1 < 0 ? 0 : (nlo[0] >> 1) - 1;
                    hi[0] = (nhi[0] >> 1) + 1 > da[0] ? da[0] : (nhi[0] >> 1) + 1;
                }
                join(need[c.srcA], lo, hi);
            }
        }
    }
    // L0 fused into L1 (conv_l0l1_fused_kernel): volume path of the split-fp16 family when L1 is a plain single-row-tile layer with one z block
    static const bool fuse01_on = !(getenv("CT_FUSE_L0L1") && atoi(getenv("CT_FUSE_L0L1")) == 0);
    bool fuse01 = false;
    if (fuse01_on && vsrc && !layer_dump && h->first_f16 && h->convs.size() > 1) {
        const ConvPlan& c0 = h->convs[0]; const ConvPlan& c1 = h->convs[1];
        const int* d1 = h->dims[c1.level];
        fuse01 = c0.cout == 8 && c1.cin == 8 && c1.bf && c1.f16 && !c1.c8 && !c1.fold && !c1.head && c1.nt_total == 1 && c1.NT == 1 && c1.srcA < 0 &&
                 c1.level == c0.level && d1[2] == 16 && d1[0] % TX == 0 && d1[1] % TY == 0 &&
                 c1.region[0] == 0 && c1.region[2] == 0 && c1.region[1] == d1[0] && c1.region[3] == d1[1] &&
                 (size_t)vsrc->q.vx * vsrc->q.vy * vsrc->q.vz * 4 < ((size_t)1 << 30);            // (its gather addresses the volume with 32-bit offsets)
    }
    h->fused01 = fuse01;
    for (size_t i = 0; i < h->convs.size(); ++i) {
        ConvPlan& c = h->convs[i];
        const int* d = h->dims[c.level];
        if (i == 0 && fuse01) continue;                        // evaluated inside layer 1's workgroups
        TimedScope timed(h, (int)i, st);
        if (i == 0) {
            const size_t nvox = (size_t)P * d[0] * d[1] * d[2];
            const unsigned nblk = (unsigned)((nvox + 255) / 256);
            const float* wt = h->d_weights + h->first_w_off;
            const float* epi = h->d_weights + c.epi_off;
            if (vsrc && c.cout == 8) {
                const int nb1 = P * ((d[0] + F1X - 1) / F1X) * ((d[1] + F1Y - 1) / F1Y) * ((d[2] + F1Z - 1) / F1Z);
                if (h->first_f16) {
                    static const int occ = getenv("CT_FIRST_OCC") ? atoi(getenv("CT_FIRST_OCC")) : 5;
                    const u32x4* wf = reinterpret_cast<const u32x4*>(h->d_weights + h->first_f16_off);
                    uint32_t* am = any_f16 ? aptr(c.dst) : nullptr;
                    if (occ >= 8) hipLaunchKernelGGL(conv_first_f16_kernel<8>, dim3(nb1), dim3(256), 0, st, vsrc->vol, vsrc->q, vsrc->p_begin, wf, h->first_wscale_inv, epi, tptr(c.dst), ad.act, am);
                    else if (occ == 5) hipLaunchKernelGGL(conv_first_f16_kernel<5>, dim3(nb1), dim3(256), 0, st, vsrc->vol, vsrc->q, vsrc->p_begin, wf, h->first_wscale_inv, epi, tptr(c.dst), ad.act, am);
                    else if (occ >= 6) hipLaunchKernelGGL(conv_first_f16_kernel<6>, dim3(nb1), dim3(256), 0, st, vsrc->vol, vsrc->q, vsrc->p_begin, wf, h->first_wscale_inv, epi, tptr(c.dst), ad.act, am);
                    else hipLaunchKernelGGL(conv_first_f16_kernel<4>, dim3(nb1), dim3(256), 0, st, vsrc->vol, vsrc->q, vsrc->p_begin, wf, h->first_wscale_inv, epi, tptr(c.dst), ad.act, am);
                } else
                hipLaunchKernelGGL(conv_first_mfma_kernel, dim3(nb1), dim3(256), 0, st, vsrc->vol, vsrc->q, vsrc->p_begin,
                                   h->d_weights + h->first_mfma_off, epi, tptr(c.dst), ad.act, any_f16 ? aptr(c.dst) : nullptr);
            } else if (c.cout == 8)
                hipLaunchKernelGGL(conv_first_kernel<8>, dim3(nblk), dim3(256), 0, st, tptr(c.srcB), wt, epi, tptr(c.dst), P, d[0], d[1], d[2], ad.act,
                                   any_f16 ? aptr(c.dst) : nullptr);
            else if (c.cout == 64)
                hipLaunchKernelGGL(conv_first_kernel<64>, dim3(nblk), dim3(256), 0, st, tptr(c.srcB), wt, epi, tptr(c.dst), P, d[0], d[1], d[2], ad.act,
                                   any_f16 ? aptr(c.dst) : nullptr);
            else return CT_ESHAPE;
            HIPCHK(hipGetLastError());
        } else {
            ConvArgs a{};
            a.srcB = tptr(c.srcB); a.CB = c.CB;
            if (c.srcA >= 0) {
                const int* da = h->dims[c.level + 1];
                a.srcA = tptr(c.srcA); a.CA = c.CA; a.AX = da[0]; a.AY = da[1]; a.AZ = da[2];
                a.ux = ad.pool[0] == 2; a.uy = ad.pool[1] == 2; a.uz = ad.pool[2] == 2;
            }
            a.X = d[0]; a.Y = d[1]; a.Z = d[2];
            a.nchunks = c.cin / 8;
            a.wpack = reinterpret_cast<const f32x4*>(h->d_weights + c.wpack_off);
            a.epi = h->d_weights + c.epi_off;
            a.out = (c.head && !layer_dump) ? nullptr : tptr(c.dst);
            a.cout = c.cout; a.nt_total = c.nt_total; a.ngroups = c.nt_total / c.NT;
            {
                static const bool lowa_on = !(getenv("CT_CONV_LOWA") && atoi(getenv("CT_CONV_LOWA")) == 0);
                a.lowa = lowa_on && c.fold && c.srcA >= 0 && a.ux == 1 && a.uy == 1 && a.uz == 0 && (d[0] & 1) == 0 && (d[1] & 1) == 0;
            }
            a.sx0 = c.store[0]; a.sx1 = c.store[1]; a.sy0 = c.store[2]; a.sy1 = c.store[3];
            if (layer_dump || !vsrc) { a.sx0 = 0; a.sx1 = d[0]; a.sy0 = 0; a.sy1 = d[1]; }
            if (c.pool_dst >= 0) {
                const int* dp = h->dims[c.level + 1];
                a.pool = tptr(c.pool_dst); a.pz = ad.pool[2]; a.PX = dp[0]; a.PY = dp[1]; a.PZ = dp[2];
            }
            if (c.head) {
                a.head = h->d_weights + h->head_off; a.head_out = prob_out;
                if (vol_out && vsrc && c.bf) {                 // split kernels: the centre crops go straight into the stitched volume
                    const TileGeom& q = vsrc->q;
                    a.vol_out = vol_out;
                    a.vb[0] = q.bx; a.vb[1] = q.by; a.vb[2] = q.bz; a.vc[0] = q.cx; a.vc[1] = q.cy; a.vc[2] = q.cz;
                    a.vv[0] = q.vx; a.vv[1] = q.vy; a.vv[2] = q.vz;
                }
            }
            a.act = ad.act;
            a.nxcd = nxcd;
#ifdef CT_TRACE
            a.trace = ((int)i == ct_trace_layer) ? ct_trace_buf : nullptr;
#endif
            if (c.f16) {
                a.amaxB = aptr(c.srcB); a.amaxA = c.srcA >= 0 ? aptr(c.srcA) : a.amaxB;
                a.amax_out = c.head ? nullptr : aptr(c.dst);
                a.wscale_inv = c.wscale_inv;
            }
            // levels with Z <= 8 (all of unet3_b, the bottom of unet3_c): 8 x 8 x 8 tiles whose MFMA columns hold two (x, y)
            // columns x 8 z instead of one x 16 z with half the lanes on padding (split-bf16 kernels, CT_CONV_Z8=0: off)
            const bool z8 = z8_on && c.bf && !c.c8 && d[2] <= 8;
            // 4 x 10 tiles for plain split-fp16 layers without a fused pool, where they cover strictly fewer columns than 4 x 8 ones (20 wide: 20
            // instead of 24 -- the bottom convs of unet3_a, measured 10-12 % faster; levels that both shapes tile exactly are 2-20 % SLOWER with
            // them: profiles/r05_conv_experiments.txt section 7).  CT_CONV_Y10 = 0: never; = a bit mask of conv indices: exactly those layers.
            static const char* y10_env = getenv("CT_CONV_Y10");
            static const unsigned y10_mask = y10_env ? (unsigned)strtoul(y10_env, nullptr, 0) : 0u;
            const int ry = c.region[3] - c.region[2];
            const bool y10_ok = c.bf && c.f16 && !c.c8 && !c.fold && !z8 && c.pool_dst < 0 && !(i == 1 && fuse01) && (c.NT <= 2 || (c.NT == 4 && !c.head));
            const bool y10 = y10_ok && (y10_env ? ((y10_mask >> i) & 1u) != 0 : (ry + 9) / 10 * 10 < (ry + TY - 1) / TY * TY);
            {
                const int tw = z8 ? 8 : TX, th = y10 ? 10 : TY;
                a.tx0 = c.region[0]; a.ty0 = c.region[2];
                a.tilesX = (c.region[1] - c.region[0] + tw - 1) / tw; a.tilesY = (c.region[3] - c.region[2] + th - 1) / th;
                a.nx0 = c.needed[0]; a.nx1 = c.needed[1]; a.ny0 = c.needed[2]; a.ny1 = c.needed[3];
                a.gx1 = a.gy1 = -1;
                const bool edges = crop_mode > 0 && vsrc && !layer_dump && c.f16 &&
                                   (c.region_e[1] < c.region[1] || c.region_e[3] < c.region[3] || c.needed_e[1] < c.needed[1] || c.needed_e[3] < c.needed[3]);
                if (edges || a.vol_out) {
                    a.p_first = vsrc->p_begin; a.pg_yz = vsrc->q.gy * vsrc->q.gz; a.pg_z = vsrc->q.gz;
                    const int dd[2] = {a.pg_yz, a.pg_z};
                    for (int k = 0; k < 2; ++k) a.mdivp[k] = dd[k] <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)dd[k]);
                }
                if (edges) {
                    a.gx1 = vsrc->q.gx - 1; a.gy1 = vsrc->q.gy - 1;
                    a.cx1e = c.region_e[1]; a.cy1e = c.region_e[3]; a.nx1e = c.needed_e[1]; a.ny1e = c.needed_e[3];
                    const int dd[2] = {a.pg_yz, a.pg_z};
                    for (int k = 0; k < 2; ++k) a.mdivp[k] = dd[k] <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)dd[k]);
                }
            }
            a.zblocks = z8 ? (d[2] + 7) / 8 : (d[2] + 15) / 16;
            // small grids: a wide layer whose NT = 4 grid is only a few "waves" of workgroups loses up to a third to the
            // tail; NT = 2 with two cout groups doubles the workgroups (and fits 3 per CU) at the price of staging twice
            int NTsel = c.NT;
            {
                const long nblk4 = (long)P * a.tilesX * a.tilesY * a.zblocks * a.ngroups;
                static const int thr = getenv("CT_CONV_SPLIT_THR") ? atoi(getenv("CT_CONV_SPLIT_THR")) : 4096;
                if (!c.c8 && !c.head && c.NT == 4 && (nblk4 < thr || y10)) { NTsel = 2; a.ngroups = c.nt_total / 2; }   // (the fused head needs all channels in one block)
            }
            c.nt_used = NTsel;
            c.tile[0] = z8 ? 8 : TX; c.tile[1] = y10 ? 10 : TY; c.tile[2] = z8 ? 8 : 16;
            int rc;
            if (i == 1 && fuse01) {
                ConvArgs af = a;
                const uint32_t nblk = (uint32_t)P * af.tilesX * af.tilesY;
                af.ngroups = 1;
                if (af.nxcd < 1) af.nxcd = 1;
                af.xper = nblk / (uint32_t)af.nxcd; af.xrem = nblk - af.xper * (uint32_t)af.nxcd;
                const int dd[5] = {1, 1, af.tilesY, af.tilesX, af.nxcd};
                for (int k = 0; k < 5; ++k) af.mdiv[k] = dd[k] <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)dd[k]);
                af.p_first = vsrc->p_begin; af.pg_yz = vsrc->q.gy * vsrc->q.gz; af.pg_z = vsrc->q.gz;     // (the kernel decodes the patch's grid place itself)
                { const int dp[2] = {af.pg_yz, af.pg_z};
                  for (int k = 0; k < 2; ++k) af.mdivp[k] = dp[k] <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint32_t)dp[k]); }
                FirstArgs fa{vsrc->vol, vsrc->q, vsrc->p_begin, reinterpret_cast<const u32x4*>(h->d_weights + h->first_f16b_off), h->first_wscale_inv,
                             h->d_weights + h->convs[0].epi_off, h->first_bound_a, h->first_bound_b};
                hipLaunchKernelGGL(conv_l0l1_fused_kernel, dim3(nblk), dim3(256), 0, st, af, fa);
                rc = (int)hipGetLastError();
            } else
            if (c.bf) {
                rc = c.f16 ? launch_conv_split<true>(a, P, NTsel, c.c8, c.fold, z8, y10, st)
                           : launch_conv_split<false>(a, P, NTsel, c.c8, c.fold, z8, false, st);
            } else if (c.c8) {
                const int nblk = P * a.tilesX * a.tilesY * a.zblocks;
                if (c.fold) hipLaunchKernelGGL(conv3_mfma_c8_fold_kernel, dim3(nblk), dim3(256), 0, st, a);
                else        hipLaunchKernelGGL(conv3_mfma_c8_kernel, dim3(nblk), dim3(256), 0, st, a);
                rc = (int)hipGetLastError();
            } else
            switch (NTsel) {
                case 1: rc = launch_conv<1>(a, P, c.fold, st); break;
                case 2: rc = launch_conv<2>(a, P, c.fold, st); break;
                case 4: rc = launch_conv<4>(a, P, c.fold, st); break;
                default: return CT_ESHAPE;
            }
            if (rc) return rc;
            if (layer_dump) {
                const float* srcp = tptr(c.dst);
                const size_t n = (size_t)d[0] * d[1] * d[2] * c.cout;
                hipLaunchKernelGGL(unblock_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                   srcp, layer_dump + dump_off, d[0], d[1], d[2], c.cout);
                HIPCHK(hipGetLastError());
                dump_off += n;
            }
            continue;
        }
        if (layer_dump) {
            const size_t n = (size_t)d[0] * d[1] * d[2] * c.cout;
            hipLaunchKernelGGL(unblock_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               tptr(c.dst), layer_dump + dump_off, d[0], d[1], d[2], c.cout);
            HIPCHK(hipGetLastError());
            dump_off += n;
        }
    }
    return CT_OK;
}

int ct_unet_predict_patches(ct_unet_t* h, const float* patches_in, int n_patches, float* prob_out,
                            void* workspace, size_t workspace_bytes, float* layer_dump, ct_stream_t stream) {
    if (!h || !patches_in || !prob_out || !workspace || n_patches <= 0) return CT_EINVAL;
    if (workspace_bytes < ct_unet_workspace_bytes(h, n_patches)) return CT_EWORKSPACE;
    DeviceGuard dg(h->device);
    if (dg.err != hipSuccess) return (int)dg.err;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t vox0 = (size_t)h->dims[0][0] * h->dims[0][1] * h->dims[0][2];
    // tensor 0 (input patches) is read in place from the caller's buffer: copy is avoided by
    // pointing tensor 0 at it when contiguous -- simplest is one D2D copy (1.6 MB / patch).
    HIPCHK(hipMemcpyAsync(ws, patches_in, vox0 * n_patches * sizeof(float), hipMemcpyDeviceToDevice, st));
    return run_network(h, ws, n_patches, prob_out, layer_dump, st);
}

static int make_geom(const int v[3], const int net[3], const int shrink[3], TileGeom& q) {
    if (!v || !net || !shrink) return CT_EINVAL;
    int* dst[5][3] = {{&q.vx, &q.vy, &q.vz}, {&q.nx, &q.ny, &q.nz}, {&q.cx, &q.cy, &q.cz}, {&q.gx, &q.gy, &q.gz}, {&q.bx, &q.by, &q.bz}};
    for (int i = 0; i < 3; ++i) {
        if (v[i] <= 0 || net[i] <= 0 || shrink[i] < 0) return CT_EINVAL;
        const int c = net[i] - 2 * shrink[i];
        if (c <= 0) return CT_ESHAPE;
        *dst[0][i] = v[i]; *dst[1][i] = net[i]; *dst[2][i] = c;
        *dst[3][i] = (v[i] + c - 1) / c;        // ceil(size / centre)   (unet3d.py:277)
        *dst[4][i] = shrink[i];
    }
    return CT_OK;
}

int ct_tile_plan(const int v[3], const int net[3], const int shrink[3], int centre[3], int grid[3]) {
    TileGeom q; int rc = make_geom(v, net, shrink, q);
    if (rc) return rc;
    if (centre) { centre[0] = q.cx; centre[1] = q.cy; centre[2] = q.cz; }
    if (grid) { grid[0] = q.gx; grid[1] = q.gy; grid[2] = q.gz; }
    return CT_OK;
}

int ct_tile_gather_reflect(const float* vol, const int v[3], const int net[3], const int shrink[3],
                           int p_begin, int n, float* patches, ct_stream_t stream) {
    TileGeom q; int rc = make_geom(v, net, shrink, q);
    if (rc) return rc;
    if (!vol || !patches || n <= 0 || p_begin < 0 || p_begin + n > q.gx * q.gy * q.gz) return CT_EINVAL;
    const size_t tot = (size_t)q.nx * q.ny * q.nz * n;
    hipLaunchKernelGGL(tile_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       vol, q, p_begin, n, patches);
    return (int)hipGetLastError();
}

int ct_tile_scatter_center(const float* pred, const int v[3], const int net[3], const int shrink[3],
                           int p_begin, int n, float* out_vol, ct_stream_t stream) {
    TileGeom q; int rc = make_geom(v, net, shrink, q);
    if (rc) return rc;
    if (!pred || !out_vol || n <= 0 || p_begin < 0 || p_begin + n > q.gx * q.gy * q.gz) return CT_EINVAL;
    const size_t tot = (size_t)q.cx * q.cy * q.cz * n;
    hipLaunchKernelGGL(tile_scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pred, q, p_begin, n, out_vol);
    return (int)hipGetLastError();
}

static int tile_crops(bool pack, float* vol, const int v[3], const int net[3], const int shrink[3], int p_begin, int n,
                      float* crops, ct_stream_t stream) {
    TileGeom q; int rc = make_geom(v, net, shrink, q);
    if (rc) return rc;
    if (!vol || !crops || n <= 0 || p_begin < 0 || p_begin + n > q.gx * q.gy * q.gz) return CT_EINVAL;
    const size_t tot = (size_t)q.cx * q.cy * q.cz * n;
    if (pack) hipLaunchKernelGGL(tile_crops_kernel<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vol, q, p_begin, n, crops);
    else hipLaunchKernelGGL(tile_crops_kernel<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vol, q, p_begin, n, crops);
    return (int)hipGetLastError();
}

int ct_tile_pack_crops(const float* vol, const int v[3], const int net[3], const int shrink[3], int p_begin, int n,
                       float* crops, ct_stream_t stream) {
    return tile_crops(true, const_cast<float*>(vol), v, net, shrink, p_begin, n, crops, stream);
}

int ct_tile_unpack_crops(const float* crops, const int v[3], const int net[3], const int shrink[3], int p_begin, int n,
                         float* out_vol, ct_stream_t stream) {
    return tile_crops(false, out_vol, v, net, shrink, p_begin, n, const_cast<float*>(crops), stream);
}

int ct_unet_predict_volume(ct_unet_t* h, const float* vol, const int v[3], const int shrink[3],
                           int p_begin, int n, float* out_vol, void* workspace, size_t workspace_bytes,
                           ct_stream_t stream) {
    if (!h || !vol || !out_vol || !workspace || n <= 0) return CT_EINVAL;
    DeviceGuard dg(h->device);
    if (dg.err != hipSuccess) return (int)dg.err;
    TileGeom q; int rc = make_geom(v, h->ad.in, shrink, q);
    if (rc) return rc;
    if (p_begin < 0 || p_begin + n > q.gx * q.gy * q.gz) return CT_EINVAL;
    const size_t per = ct_unet_workspace_bytes(h, 1) - 256;
    if (workspace_bytes < per + 256) return CT_EWORKSPACE;
    const int batch_cap = (int)((workspace_bytes - 256) / per);
    float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t vox0 = (size_t)q.nx * q.ny * q.nz;
    hipStream_t st = (hipStream_t)stream;
    for (int done = 0; done < n;) {
        const int nb = (n - done) < batch_cap ? (n - done) : batch_cap;
        float* prob = ws + h->floats_per_patch * (size_t)nb + amax_words(h, nb);          // [nb][X][Y][Z]
        const bool fused_first = h->convs[0].cout == 8 && !getenv("CT_NO_FUSED_FIRST");
        if (!fused_first) {
            rc = ct_tile_gather_reflect(vol, v, h->ad.in, shrink, p_begin + done, nb, ws, stream);   // tensor 0
            if (rc) return rc;
        }
        VolSource vs{vol, q, p_begin + done};
        // split-kernel head layers write their centre crops straight into the stitched volume (CT_DIRECT_STITCH=0: per-patch maps + scatter)
        static const bool direct_on = !(getenv("CT_DIRECT_STITCH") && atoi(getenv("CT_DIRECT_STITCH")) == 0);
        const bool direct = direct_on && fused_first && h->convs.back().head && h->convs.back().bf;
        rc = run_network(h, ws, nb, prob, nullptr, st, fused_first ? &vs : nullptr, direct ? out_vol : nullptr);
        if (rc) return rc;
        if (!direct) {
            rc = ct_tile_scatter_center(prob, v, h->ad.in, shrink, p_begin + done, nb, out_vol, stream);
            if (rc) return rc;
        }
        (void)vox0;
        done += nb;
    }
    return CT_OK;
}

}  // extern "C"
