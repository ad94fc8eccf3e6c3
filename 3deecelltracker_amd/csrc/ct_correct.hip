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

#include "../../include/ctamd.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct CorrGeom { int X, Y, Z, factor, zstart; };        // original grid, z interpolation factor, first original slice

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

// movement of every cell on the interpolated grid: round((coords - vol1) * (1, 1, factor)), numpy half-to-even
__global__ void movements_kernel(const float* __restrict__ coords, const float* __restrict__ vol1, int n, int factor,
                                 int32_t* __restrict__ mov) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const float d = coords[i] - vol1[i];                       // Coordinates.__sub__ : float32 raw difference
    const double s = (i % 3 == 2) ? (double)factor : 1.0;
    mov[i] = (int32_t)rint((double)d * s);
}

// visit the voxels of cell i's moved mask that land on an original z slice inside the image
template <typename F>
__device__ __forceinline__ void for_cell_voxels(const CorrGeom g, const int32_t* bbox, const uint8_t* sub, const int32_t* mov,
                                                int* err, F&& f) {
    const int bx = bbox[0], by = bbox[1], bz = bbox[2], sx = bbox[3], sy = bbox[4], sz = bbox[5];
    const int ox = bx + mov[0], oy = by + mov[1], oz = bz + mov[2];
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
                                                             const int32_t* __restrict__ mov, unsigned int* __restrict__ cnt, int* __restrict__ err) {
    const int i = blockIdx.x;
    if (missed[i]) return;
    for_cell_voxels(g, bbox + 6 * i, subs + offs[i], mov + 3 * i, err,
                    [&](int x, int y, int k) { atomicAdd(&cnt[((size_t)x * g.Y + y) * g.Z + k], 1u); });
}

// per-cell centre of mass over its non-overlapping voxels; lost cells keep round(coords); writes the signed max of
// round(delta * (1,1,factor)) into flag[0] (offset by 2^30 so that atomicMax on unsigned works)
__global__ __launch_bounds__(256) void centre_of_mass_kernel(CorrGeom g, const float* __restrict__ prob, const int32_t* __restrict__ bbox,
                                                             const uint8_t* __restrict__ subs, const long long* __restrict__ offs,
                                                             const uint8_t* __restrict__ missed, const int32_t* __restrict__ mov,
                                                             const unsigned int* __restrict__ cnt, float* __restrict__ coords,
                                                             unsigned int* __restrict__ flag, int* __restrict__ err) {
    __shared__ double red[4][4];
    const int i = blockIdx.x;
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
            for (int d = 0; d < 3; ++d) nw[d] = (float)(int)rintf(coords[3 * i + d]);
        } else { nw[0] = (float)cx; nw[1] = (float)(t[2] / t[0]); nw[2] = (float)(t[3] / t[0]); }
        int mx = -(1 << 29);
        for (int d = 0; d < 3; ++d) {
            const float delta = nw[d] - coords[3 * i + d];
            const int di = (int)rint((double)delta * (d == 2 ? (double)g.factor : 1.0));
            mx = max(mx, di);
            coords[3 * i + d] = nw[d];
        }
        atomicMax(flag, (unsigned int)(mx + (1 << 30)));
    }
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
    int it = 0;
    for (it = 1; it <= max_repetition; ++it) {
        HIPCHK(hipMemsetAsync(cnt, 0, nvox * 4, st));
        HIPCHK(hipMemsetAsync(flag, 0, 128, st));
        hipLaunchKernelGGL(movements_kernel, dim3((3 * n_cells + 255) / 256), dim3(256), 0, st, coords_raw, coord_vol1_raw, n_cells, factor, mov);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(scatter_counts_kernel, dim3(n_cells), dim3(256), 0, st, g, bbox, subimages, sub_offsets, missed, mov, cnt, err);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(centre_of_mass_kernel, dim3(n_cells), dim3(256), 0, st, g, prob, bbox, subimages, sub_offsets, missed, mov, cnt,
                           coords_raw, flag, err);
        LAUNCH_CHECK();
        unsigned int h[32];
        HIPCHK(hipMemcpyAsync(h, flag, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (((int*)h)[16]) return CT_ESHAPE;                           // a moved bounding box left the image (reference: ValueError)
        const int mx = (int)h[0] - (1 << 30);
        if (mx < 1) break;                                             // np.max(delta.interp) < 0.5  (signed max, as in the reference)
    }
    if (iterations) *iterations = it > max_repetition ? max_repetition : it;
    return CT_OK;
}

}  // extern "C"
