// How fast can a sweep over 8.4 M int32 labels (2 % non-zero, in runs) be, and what does the histogram's bookkeeping add?  Variants of
// ws_bincount_kernel timed with HIP events.  (scripts/probe: stand-alone, not part of the library.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_onethread(long long V, const int* __restrict__ labels, int K, unsigned int* __restrict__ counts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int lab = 0;
    if (i < V) lab = labels[i];
    const bool active = lab > 0 && lab <= K;
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int ll = __shfl(lab, leader);
        const unsigned long long same = __ballot(active && lab == ll) & todo;
        if (lane == leader) atomicAdd(&counts[ll], (unsigned int)__popcll(same));
        todo &= ~same;
    }
}
__global__ void k_loadonly(long long V, const int* __restrict__ labels, unsigned int* __restrict__ counts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int lab = 0;
    if (i < V) lab = labels[i];
    if (lab == 0x7fffffff) counts[0] = 1;
}
__global__ void k_vec4(long long V, const int* __restrict__ labels, int K, unsigned int* __restrict__ counts) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * t >= V) return;
    const int4 v = reinterpret_cast<const int4*>(labels)[t];
    if ((v.x | v.y | v.z | v.w) == 0) return;
    const int a[4] = {v.x, v.y, v.z, v.w};
    int prev = 0; unsigned int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lab = (a[k] > 0 && a[k] <= K) ? a[k] : 0;
        if (lab == prev) ++cnt; else { if (cnt && prev) atomicAdd(&counts[prev], cnt); prev = lab; cnt = 1; }
    }
    if (cnt && prev) atomicAdd(&counts[prev], cnt);
}
template <int PER>
__global__ void k_vec4_loop(long long V, const int* __restrict__ labels, int K, unsigned int* __restrict__ counts) {
    // a workgroup owns PER * 1024 consecutive voxels: PER int4 loads per thread, issued together
    const long long base = ((long long)blockIdx.x * PER) * 256 + threadIdx.x;
    int4 v[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) { const long long t = base + (long long)p * 256; v[p] = 4 * t < V ? reinterpret_cast<const int4*>(labels)[t] : int4{0, 0, 0, 0}; }
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        if ((v[p].x | v[p].y | v[p].z | v[p].w) == 0) continue;
        const int a[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
        int prev = 0; unsigned int cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lab = (a[k] > 0 && a[k] <= K) ? a[k] : 0;
            if (lab == prev) ++cnt; else { if (cnt && prev) atomicAdd(&counts[prev], cnt); prev = lab; cnt = 1; }
        }
        if (cnt && prev) atomicAdd(&counts[prev], cnt);
    }
}
__global__ void k_stride(long long V, const int* __restrict__ labels, int K, unsigned int* __restrict__ counts) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; 4 * t < V; t += (long long)gridDim.x * blockDim.x) {
        const int4 v = reinterpret_cast<const int4*>(labels)[t];
        if ((v.x | v.y | v.z | v.w) == 0) continue;
        const int a[4] = {v.x, v.y, v.z, v.w};
        int prev = 0; unsigned int cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lab = (a[k] > 0 && a[k] <= K) ? a[k] : 0;
            if (lab == prev) ++cnt; else { if (cnt && prev) atomicAdd(&counts[prev], cnt); prev = lab; cnt = 1; }
        }
        if (cnt && prev) atomicAdd(&counts[prev], cnt);
    }
}
__global__ void k_copy4(long long V, const int* __restrict__ in, int* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * t < V) reinterpret_cast<int4*>(out)[t] = reinterpret_cast<const int4*>(in)[t];
}
__global__ void k_copy1(long long V, const int* __restrict__ in, int* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < V) out[t] = in[t];
}

int main() {
    const int X = 512, Y = 512, Z = 32; const long long V = (long long)X * Y * Z; const int K = 8192;
    std::vector<int> h(V, 0);
    // ~600 blobs of ~280 voxels: 6 x 6 columns x 8 z
    unsigned s = 12345; int lab = 0;
    for (int b = 0; b < 600; ++b) {
        s = s * 1664525u + 1013904223u; const int x0 = (s >> 8) % (X - 8);
        s = s * 1664525u + 1013904223u; const int y0 = (s >> 8) % (Y - 8);
        s = s * 1664525u + 1013904223u; const int z0 = (s >> 8) % (Z - 8);
        ++lab;
        for (int x = 0; x < 6; ++x) for (int y = 0; y < 6; ++y) for (int z = 0; z < 8; ++z) h[((long long)(x0 + x) * Y + (y0 + y)) * Z + z0 + z] = lab;
    }
    int *d, *o; unsigned int* c;
    CK(hipMalloc(&d, V * 4)); CK(hipMalloc(&o, V * 4)); CK(hipMalloc(&c, (K + 1) * 4));
    CK(hipMemcpy(d, h.data(), V * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        (void)hipDeviceSynchronize();
        float best = 1e9f, tot = 0.f;
        for (int r = 0; r < 20; ++r) {
            (void)hipMemsetAsync(c, 0, (K + 1) * 4, 0);
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot += ms;
        }
        printf("%-44s best %7.1f us  mean %7.1f us\n", name, best * 1e3f, tot / 20 * 1e3f);
    };
    const unsigned nb = (unsigned)((V + 255) / 256), nb4 = (unsigned)((V / 4 + 255) / 256);
    run("1 voxel/thread, wave-aggregated (library)", [&] { k_onethread<<<nb, 256>>>(V, d, K, c); });
    run("1 voxel/thread, 1024-thread groups", [&] { k_onethread<<<(unsigned)((V + 1023) / 1024), 1024>>>(V, d, K, c); });
    run("1 voxel/thread, load only", [&] { k_loadonly<<<nb, 256>>>(V, d, c); });
    run("int4/thread, run-aggregated", [&] { k_vec4<<<nb4, 256>>>(V, d, K, c); });
    run("2 x int4/thread", [&] { k_vec4_loop<2><<<(nb4 + 1) / 2, 256>>>(V, d, K, c); });
    run("4 x int4/thread", [&] { k_vec4_loop<4><<<(nb4 + 3) / 4, 256>>>(V, d, K, c); });
    run("8 x int4/thread", [&] { k_vec4_loop<8><<<(nb4 + 7) / 8, 256>>>(V, d, K, c); });
    run("grid-stride int4, 2048 groups", [&] { k_stride<<<2048, 256>>>(V, d, K, c); });
    run("grid-stride int4, 1024 groups", [&] { k_stride<<<1024, 256>>>(V, d, K, c); });
    run("copy int (1/thread)", [&] { k_copy1<<<nb, 256>>>(V, d, o); });
    run("copy int4", [&] { k_copy4<<<nb4, 256>>>(V, d, o); });
    std::vector<unsigned int> hc(K + 1);
    k_vec4_loop<4><<<(nb4 + 3) / 4, 256>>>(V, d, K, c);
    CK(hipMemcpy(hc.data(), c, (K + 1) * 4, hipMemcpyDeviceToHost));
    printf("check: counts[1] = %u (accumulated over the timed runs)\n", hc[1]);
    return 0;
}
