// ct_correct.hip -- accurate correction of cell centres on the probability map.
//
// What it replaces (reference CellTracker/coord_image_transformer.py):
//   :292-369 move_cells          moved per-cell masks summed into a label image + an overlap-count image (z-interpolated grid)
//   :449-489 _correction_once    labels[overlap > 1] = 0; scipy.ndimage.center_of_mass(prob, labels, 1..n); lost cells keep
//                                their rounded position; delta = new - old
//   :406-447 accurate_correction repeat <= max_repetition times until max(delta.interp) < 0.5
//
// The label image is never built: only the overlap counts on the original z slices are needed (a voxel belongs to cell i
// iff cell i's moved mask covers it and the count is 1), so one iteration = clear counts, scatter (one block per cell,
// atomics), per-cell fp64 reduction of prob * (x, y, k).  Coordinates stay float32 "raw" like the reference's type.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/ctamd.h"
#include "ct_fill.h"
#include "ct_fresh.h"


// (control words written by one launch and read by a later one at wave-uniform addresses go through fresh_f32 / fresh_i32: ct_fresh.h)

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct CorrGeom { int X, Y, Z, factor, zstart; };        // original grid, z interpolation factor, first original slice

// Sum over the wave (read through lane 63): DPP steps on the two halves of the double -- the same pairs in the same order as an xor
// butterfly (quad 1, quad 2, half-row mirror, row mirror, row 15 -> next row, lane 31 -> upper half; a + b == b + a, so the same bits)
// without the butterfly's twelve ds_bpermute round trips.
__device__ __forceinline__ double wave_sum(double v) {
#define CT_DPP_ADD64(ctrl, rmask) { int lo = __double2loint(v), hi = __double2hiint(v);                                              \
                                    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, rmask, 0xf, false);                                  \
                                    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, rmask, 0xf, false); v += __hiloint2double(hi, lo); }
    CT_DPP_ADD64(0xB1, 0xf) CT_DPP_ADD64(0x4E, 0xf) CT_DPP_ADD64(0x141, 0xf) CT_DPP_ADD64(0x140, 0xf) CT_DPP_ADD64(0x142, 0xa) CT_DPP_ADD64(0x143, 0xc)
#undef CT_DPP_ADD64
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// movement of every cell on the interpolated grid: round((coords - vol1) * (1, 1, factor)), numpy half-to-even
// (`done` != 0: an earlier round already met the stopping rule -- rounds enqueued ahead of the host's look at the flags do nothing)
__global__ void movements_kernel(const float* __restrict__ coords, const float* __restrict__ vol1, int n, int factor,
                                 int32_t* __restrict__ mov, const int* __restrict__ done) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n || fresh_i32(done)) return;
    const float d = coords[i] - vol1[i];                       // Coordinates.__sub__ : float32 raw difference
    const double s = (i % 3 == 2) ? (double)factor : 1.0;
    mov[i] = (int32_t)rint((double)d * s);
}

// visit the voxels of cell i's moved mask that land on an original z slice inside the image
template <typename F>
__device__ __forceinline__ void for_cell_voxels(const CorrGeom g, const int32_t* bbox, const uint8_t* sub, const int32_t* mov,
                                                int* err, F&& f) {
    const int bx = bbox[0], by = bbox[1], bz = bbox[2], sx = bbox[3], sy = bbox[4], sz = bbox[5];
    const int ox = bx + fresh_i32(mov), oy = by + fresh_i32(mov + 1), oz = bz + fresh_i32(mov + 2);      // (rewritten every round: see fresh_i32)
    const int ZI = g.Z * g.factor;
    // clipped ranges (reference raises ValueError when a clipped range is empty)
    if (max(ox, 0) >= min(ox + sx, g.X) || max(oy, 0) >= min(oy + sy, g.Y) || max(oz, 0) >= min(oz + sz, ZI)) {
        if (threadIdx.x == 0) atomicExch(err, 1);
        return;
    }
    const int nvox = sx * sy * sz;
    for (int v = threadIdx.x; v < nvox; v += blockDim.x) {
        if (!sub[v]) continue;
        const int vz = v % sz, vy = (v / sz) % sy, vx = v / (sz * sy);
        const int x = ox + vx, y = oy + vy, zi = oz + vz;
        if (x < 0 || x >= g.X || y < 0 || y >= g.Y || zi < 0 || zi >= ZI) continue;
        const int dz = zi - g.zstart;
        if (dz < 0 || dz % g.factor) continue;
        const int k = dz / g.factor;
        if (k >= g.Z) continue;
        f(x, y, k);
    }
}

__global__ __launch_bounds__(256) void scatter_counts_kernel(CorrGeom g, const int32_t* __restrict__ bbox, const uint8_t* __restrict__ subs,
                                                             const long long* __restrict__ offs, const uint8_t* __restrict__ missed,
                                                             const int32_t* __restrict__ mov, unsigned int* __restrict__ cnt, int* __restrict__ err,
                                                             const int* __restrict__ done) {
    const int i = blockIdx.x;
    if (missed[i] || fresh_i32(done)) return;
    for_cell_voxels(g, bbox + 6 * i, subs + offs[i], mov + 3 * i, err,
                    [&](int x, int y, int k) { atomicAdd(&cnt[((size_t)x * g.Y + y) * g.Z + k], 1u); });
}

// per-cell centre of mass over its non-overlapping voxels; lost cells keep round(coords); writes the signed max of
// round(delta * (1,1,factor)) into flag[0] (offset by 2^30 so that atomicMax on unsigned works)
__global__ __launch_bounds__(256) void centre_of_mass_kernel(CorrGeom g, const float* __restrict__ prob, const int32_t* __restrict__ bbox,
                                                             const uint8_t* __restrict__ subs, const long long* __restrict__ offs,
                                                             const uint8_t* __restrict__ missed, const int32_t* __restrict__ mov,
                                                             const unsigned int* __restrict__ cnt, float* __restrict__ coords,
                                                             unsigned int* __restrict__ flag, int* __restrict__ err, const int* __restrict__ done) {
    __shared__ double red[4][4];
    const int i = blockIdx.x;
    if (fresh_i32(done)) return;
    double sw = 0.0, swx = 0.0, swy = 0.0, swz = 0.0;
    if (!missed[i])
        for_cell_voxels(g, bbox + 6 * i, subs + offs[i], mov + 3 * i, err, [&](int x, int y, int k) {
            const size_t idx = ((size_t)x * g.Y + y) * g.Z + k;
            if (cnt[idx] == 1u) {
                const double p = (double)prob[idx];
                sw += p; swx += p * x; swy += p * y; swz += p * k;
            }
        });
    sw = wave_sum(sw); swx = wave_sum(swx); swy = wave_sum(swy); swz = wave_sum(swz);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = sw; red[wv][1] = swx; red[wv][2] = swy; red[wv][3] = swz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4];
        for (int q = 0; q < 4; ++q) t[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        float nw[3];
        const double cx = t[1] / t[0];
        if (cx != cx || t[0] == 0.0) {            // 0/0 -> NaN in the reference: "lost" cell keeps its rounded position
            for (int d = 0; d < 3; ++d) nw[d] = (float)(int)rintf(fresh_f32(&coords[3 * i + d]));
        } else { nw[0] = (float)cx; nw[1] = (float)(t[2] / t[0]); nw[2] = (float)(t[3] / t[0]); }
        int mx = -(1 << 29);
        for (int d = 0; d < 3; ++d) {
            const float oldc = fresh_f32(&coords[3 * i + d]);
            const float delta = nw[d] - oldc;
            const int di = (int)rint((double)delta * (d == 2 ? (double)g.factor : 1.0));
            mx = max(mx, di);
            coords[3 * i + d] = nw[d];
        }
        atomicMax(flag, (unsigned int)(mx + (1 << 30)));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Legacy dialect: Tracker._accurate_correction (reference CellTracker/tracker.py:1177-1191) with
//   _transform_cells_quick  :1350-1389   labels moved by integer displacements on the interpolated grid inside an image
//                                        padded by the largest sub-region; a cell whose moved box leaves the padded
//                                        image is skipped (shape mismatch -> continue); overlap counts as above
//   _correction_once_interp :1310-1348   labels[overlap > 1] = 0, boundary cells = 0, centre of mass of
//                                        (image_cell_bg + image_gcn) per label, lost cells (NaN) get correction 0
//   _evaluate_correction    :1402-1413   stop when max |correction| (interpolated units) < 0.5
// State per cell: integer displacement from volume 1 (interpolated grid) and the real displacement.
// ------------------------------------------------------------------------------------------------------------------
struct LegacyGeom { int X, Y, Z, zs, ZI, padx, pady, padz; };

template <typename F>
__device__ __forceinline__ bool for_cell_voxels_legacy(const LegacyGeom g, const int32_t* bbox, const uint8_t* sub, const int32_t* disp, F&& f) {
    const int sx = bbox[3], sy = bbox[4], sz = bbox[5];
    const int ox = bbox[0] + fresh_i32(disp), oy = bbox[1] + fresh_i32(disp + 1), oz = bbox[2] + fresh_i32(disp + 2);   // (rewritten every other round)
    // padded-image slice [o + pad, o + pad + s) must lie inside [0, dim + 2 pad): otherwise numpy returns a slice of another
    // shape and the reference skips the cell (a start below zero would wrap around in numpy; treated as skipped as well)
    if (ox + g.padx < 0 || ox + sx > g.X + g.padx || oy + g.pady < 0 || oy + sy > g.Y + g.pady ||
        oz + g.padz < 0 || oz + sz > g.ZI + g.padz) return false;
    const int nvox = sx * sy * sz;
    for (int v = threadIdx.x; v < nvox; v += blockDim.x) {
        if (!sub[v]) continue;
        const int vz = v % sz, vy = (v / sz) % sy, vx = v / (sz * sy);
        const int x = ox + vx, y = oy + vy, zi = oz + vz;
        if (x < 0 || x >= g.X || y < 0 || y >= g.Y || zi < 0 || zi >= g.ZI) continue;
        const int dz = zi - g.zs / 2;                              // slices zs//2 : Z*zs : zs
        if (dz < 0 || dz % g.zs) continue;
        const int k = dz / g.zs;
        if (k >= g.Z) continue;
        f(x, y, k);
    }
    return true;
}

// i_disp = rint(r_disp * (1, 1, zs / ratio))   (_transform_real_to_interpolated :563-565)
__global__ void legacy_to_interp_kernel(const double* __restrict__ r_disp, int n, double zfac, int32_t* __restrict__ i_disp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const double v = (i % 3 == 2) ? r_disp[i] * zfac : r_disp[i];
    i_disp[i] = (int32_t)rint(v);
}

__global__ __launch_bounds__(256) void legacy_scatter_kernel(LegacyGeom g, const int32_t* __restrict__ bbox, const uint8_t* __restrict__ subs,
                                                             const long long* __restrict__ offs, const int32_t* __restrict__ disp,
                                                             unsigned int* __restrict__ cnt) {
    const int i = blockIdx.x;
    for_cell_voxels_legacy(g, bbox + 6 * i, subs + offs[i], disp + 3 * i,
                           [&](int x, int y, int k) { atomicAdd(&cnt[((size_t)x * g.Y + y) * g.Z + k], 1u); });
}

template <typename RAW>
__global__ __launch_bounds__(256) void legacy_com_kernel(LegacyGeom g, double ratio, const float* __restrict__ prob, const RAW* __restrict__ raw,
                                                         const int32_t* __restrict__ bbox, const uint8_t* __restrict__ subs,
                                                         const long long* __restrict__ offs, const uint8_t* __restrict__ on_boundary,
                                                         const unsigned int* __restrict__ cnt, const double* __restrict__ t0,
                                                         const int32_t* __restrict__ disp_in, int32_t* __restrict__ disp_out,
                                                         double* __restrict__ r_disp, unsigned int* __restrict__ flag) {
    __shared__ double red[4][4];
    const int i = blockIdx.x;
    double sw = 0.0, swx = 0.0, swy = 0.0, swz = 0.0;
    if (!on_boundary[i])
        for_cell_voxels_legacy(g, bbox + 6 * i, subs + offs[i], disp_in + 3 * i, [&](int x, int y, int k) {
            const size_t idx = ((size_t)x * g.Y + y) * g.Z + k;
            if (cnt[idx] == 1u) {
                double w = (double)prob[idx];
                if (raw) w += (double)raw[idx] / 65536.0;          // image_cell_bg + image_gcn  (:635, :1332)
                sw += w; swx += w * x; swy += w * y; swz += w * k;
            }
        });
    sw = wave_sum(sw); swx = wave_sum(swx); swy = wave_sum(swy); swz = wave_sum(swz);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = sw; red[wv][1] = swx; red[wv][2] = swy; red[wv][3] = swz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4];
        for (int q = 0; q < 4; ++q) t[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        const double zs = (double)g.zs;
        const double inv_ratio = 1.0 / ratio, inv_zs = 1.0 / zs, r_over_zs = ratio / zs, zs_over_r = zs / ratio;
        double com[3] = {t[1] / t[0], t[2] / t[0], t[3] / t[0]};
        const bool lost = (com[0] != com[0]);
        bool big = false;
        for (int d = 0; d < 3; ++d) {
            const double id = (double)fresh_i32(&disp_in[3 * i + d]);
            const double lc = (d == 2) ? t0[3 * i + 2] * inv_ratio + id * inv_zs : t0[3 * i + d] + id;
            double corr = lost ? 0.0 : com[d] - lc;
            if (d == 2) corr = corr * ratio;
            const double rd = ((d == 2) ? id * r_over_zs : id) + corr;
            r_disp[3 * i + d] = rd;
            disp_out[3 * i + d] = (int32_t)rint((d == 2) ? rd * zs_over_r : rd);
            const double test = (d == 2) ? corr * zs_over_r : corr;
            if (fabs(test) >= 0.5) big = true;
        }
        if (big) atomicOr(flag, 1u);
    }
}

// closes round `it`: np.max(delta.interp) < 0.5 (the flag holds the signed maximum + 2^30) -> done = it; the round's flag words are kept for the
// host in hist[it & 1] (it looks every second round)
__global__ void correction_round_end_kernel(const unsigned int* __restrict__ flag, int it, int* __restrict__ done, unsigned int* __restrict__ hist) {
    if (fresh_i32(done)) return;
    hist[2 * (it & 1)] = flag[0]; hist[2 * (it & 1) + 1] = flag[16];
    if ((int)flag[0] - (1 << 30) < 1 || flag[16]) *done = it;                   // (an out-of-image box also ends the loop: the host reports it)
}

}  // namespace

extern "C" {

size_t ct_correction_workspace_bytes(const int dims[3], int n_cells) {
    if (!dims || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0 || n_cells <= 0) return 0;
    return align_up((size_t)dims[0] * dims[1] * dims[2] * 4, 256) + align_up((size_t)n_cells * 12, 256) + 1024;
}

int ct_accurate_correction(const float* prob, const int dims[3], int factor, int n_cells, const int32_t* bbox, const uint8_t* subimages,
                           const long long* sub_offsets, const uint8_t* missed, const float* coord_vol1_raw, float* coords_raw,
                           int max_repetition, int* iterations, void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!prob || !dims || !bbox || !subimages || !sub_offsets || !missed || !coord_vol1_raw || !coords_raw || !workspace ||
        factor <= 0 || n_cells <= 0 || max_repetition <= 0) return CT_EINVAL;
    if (workspace_bytes < ct_correction_workspace_bytes(dims, n_cells)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    CorrGeom g{dims[0], dims[1], dims[2], factor, factor / 2};            // z_slice_original_labels = slice(f//2, f*Z, f)
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t nvox = (size_t)dims[0] * dims[1] * dims[2];
    unsigned int* cnt = (unsigned int*)ws; ws += align_up(nvox * 4, 256);
    int32_t* mov = (int32_t*)ws; ws += align_up((size_t)n_cells * 12, 256);
    unsigned int* flag = (unsigned int*)ws; int* err = (int*)(ws + 64);
    int* done = (int*)(ws + 256); unsigned int* hist = (unsigned int*)(ws + 320);      // (behind the per-round flag block)
    HIPCHK(ct_fill_async(done, 0, 128, st));
    // Rounds are enqueued two at a time: the device closes each round itself (correction_round_end_kernel) and a round enqueued after the
    // stopping rule was met does nothing, so the host waits once per two rounds instead of once per round (a frame's four rounds: two idle gaps
    // instead of four).  Same rounds, same results, same iteration count.
    int it = 0, finished = 0;
    for (it = 1; it <= max_repetition && !finished; ) {
        const int first = it;
        static const int per_sync = (getenv("CT_CORR_PAIR") && atoi(getenv("CT_CORR_PAIR")) == 0) ? 1 : 2;
        for (int k = 0; k < per_sync && it <= max_repetition; ++k, ++it) {
            HIPCHK(ct_fill_async(cnt, 0, nvox * 4, st));
            HIPCHK(ct_fill_async(flag, 0, 128, st));
            hipLaunchKernelGGL(movements_kernel, dim3((3 * n_cells + 255) / 256), dim3(256), 0, st, coords_raw, coord_vol1_raw, n_cells, factor, mov, done);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(scatter_counts_kernel, dim3(n_cells), dim3(256), 0, st, g, bbox, subimages, sub_offsets, missed, mov, cnt, err, done);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(centre_of_mass_kernel, dim3(n_cells), dim3(256), 0, st, g, prob, bbox, subimages, sub_offsets, missed, mov, cnt,
                               coords_raw, flag, err, done);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(correction_round_end_kernel, dim3(1), dim3(1), 0, st, flag, it, done, hist);
            LAUNCH_CHECK();
            static const bool trace = getenv("CT_CORR_TRACE") != nullptr;      // debugging aid: hashes of the round's state on stderr (synchronises)
            if (trace) {
                float* hc = (float*)malloc((size_t)n_cells * 12); int32_t* hm = (int32_t*)malloc((size_t)n_cells * 12);
                unsigned int* hcnt = (unsigned int*)malloc(nvox * 4); unsigned int hf[32];
                HIPCHK(hipMemcpyAsync(hc, coords_raw, (size_t)n_cells * 12, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(hm, mov, (size_t)n_cells * 12, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(hcnt, cnt, nvox * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(hf, flag, 128, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                auto fnv = [](const void* p, size_t n) { unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= ((const unsigned char*)p)[i]; h *= 1099511628211ull; } return h; };
                fprintf(stderr, "[corr] round %d flag %d err %u coords %016llx mov %016llx cnt %016llx\n", it, (int)hf[0] - (1 << 30), hf[16],
                        fnv(hc, (size_t)n_cells * 12), fnv(hm, (size_t)n_cells * 12), fnv(hcnt, nvox * 4));
                free(hc); free(hm); free(hcnt);
            }
        }
        unsigned int h[32];                                            // done | ... | hist[4] at + 16 words
        HIPCHK(hipMemcpyAsync(h, done, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const int d = (int)h[0];
        if (getenv("CT_DEBUG"))
            fprintf(stderr, "[ct_accurate_correction] rounds %d..%d: done %d, flags (max delta + 2^30, err) even round %d %u, odd round %d %u\n", first, it - 1, d,
                    (int)h[16] - (1 << 30), h[17], (int)h[18] - (1 << 30), h[19]);
        if (d) {
            if (h[16 + 2 * (d & 1) + 1]) return CT_ESHAPE;             // a moved bounding box left the image (reference: ValueError)
            finished = d;
        } else (void)first;
    }
    if (iterations) *iterations = finished ? finished : max_repetition;
    return CT_OK;
}

size_t ct_correction_legacy_workspace_bytes(const int dims[3], int n_cells) {
    if (!dims || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0 || n_cells <= 0) return 0;
    return align_up((size_t)dims[0] * dims[1] * dims[2] * 4, 256) + 2 * align_up((size_t)n_cells * 12, 256) + 1024;
}

int ct_accurate_correction_legacy(const float* prob, const void* raw, int raw_dtype, const int dims[3], int z_scaling, int interp_depth,
                                  double z_xy_ratio, int n_cells, const int32_t* bbox, const uint8_t* subimages,
                                  const long long* sub_offsets, const int pad_xyz[3], const uint8_t* on_boundary,
                                  const double* tracked_t0, double* r_disp, int32_t* i_disp, int max_repetition, int* iterations,
                                  void* workspace, size_t workspace_bytes, ct_stream_t stream) {
    if (!prob || !dims || !bbox || !subimages || !sub_offsets || !pad_xyz || !on_boundary || !tracked_t0 || !r_disp || !i_disp ||
        !workspace || z_scaling <= 0 || interp_depth <= 0 || !(z_xy_ratio > 0.0) || n_cells <= 0 || max_repetition <= 0 ||
        (raw && raw_dtype != 0 && raw_dtype != 1)) return CT_EINVAL;
    if (workspace_bytes < ct_correction_legacy_workspace_bytes(dims, n_cells)) return CT_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    LegacyGeom g{dims[0], dims[1], dims[2], z_scaling, interp_depth, pad_xyz[0], pad_xyz[1], pad_xyz[2]};
    unsigned char* ws = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t nvox = (size_t)dims[0] * dims[1] * dims[2];
    unsigned int* cnt = (unsigned int*)ws; ws += align_up(nvox * 4, 256);
    int32_t* da = (int32_t*)ws; ws += align_up((size_t)n_cells * 12, 256);
    int32_t* db = (int32_t*)ws; ws += align_up((size_t)n_cells * 12, 256);
    unsigned int* flag = (unsigned int*)ws;
    hipLaunchKernelGGL(legacy_to_interp_kernel, dim3((3 * n_cells + 255) / 256), dim3(256), 0, st, r_disp, n_cells,
                       (double)z_scaling / z_xy_ratio, da);
    LAUNCH_CHECK();
    int it = 0;
    for (it = 1; it <= max_repetition; ++it) {
        HIPCHK(ct_fill_async(cnt, 0, nvox * 4, st));
        HIPCHK(ct_fill_async(flag, 0, 64, st));
        hipLaunchKernelGGL(legacy_scatter_kernel, dim3(n_cells), dim3(256), 0, st, g, bbox, subimages, sub_offsets, da, cnt);
        LAUNCH_CHECK();
        if (raw && raw_dtype == 1)
            hipLaunchKernelGGL(legacy_com_kernel<float>, dim3(n_cells), dim3(256), 0, st, g, z_xy_ratio, prob, (const float*)raw, bbox, subimages,
                               sub_offsets, on_boundary, cnt, tracked_t0, da, db, r_disp, flag);
        else
            hipLaunchKernelGGL(legacy_com_kernel<uint16_t>, dim3(n_cells), dim3(256), 0, st, g, z_xy_ratio, prob, (const uint16_t*)raw, bbox,
                               subimages, sub_offsets, on_boundary, cnt, tracked_t0, da, db, r_disp, flag);
        LAUNCH_CHECK();
        int32_t* t = da; da = db; db = t;
        unsigned int h = 0;
        HIPCHK(hipMemcpyAsync(&h, flag, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!h) break;                                                  // _evaluate_correction: every |correction| < 0.5
    }
    HIPCHK(hipMemcpyAsync(i_disp, da, (size_t)n_cells * 12, hipMemcpyDeviceToDevice, st));
    if (iterations) *iterations = it > max_repetition ? max_repetition : it;
    return CT_OK;
}

}  // extern "C"
