// Aggressor candidates for the co-residency hazard that contain NO arithmetic at all (scripts/probe/hazard_storm.sh):
// the bisect of the split conv kernels (hazard_bisect.sh) ended at a kernel that returns at its first statement and still corrupts
// the packed-fp32 victim of another process, so what is left is the launch pattern itself.
//   wave_storm <seconds> <mode> [workgroups] [threads]
//     mode 0: empty kernel (every wave retires at once)             mode 1: every thread touches a 36-KB static LDS array
//     mode 2: empty kernel that first sleeps ~20 us                  mode 3: one barrier, then exit
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/wave_storm.hip -o scripts/probe/wave_storm
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) p[0] = 1; }
__global__ void k_lds(int* p) {
    __shared__ int big[9216];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (p && big[(threadIdx.x + 1) & 255] == 123456) p[0] = 1;
}
__global__ void k_sleep(int* p) {
    for (int i = 0; i < 300; ++i) __builtin_amdgcn_s_sleep(127);
    if (p && threadIdx.x == 12345) p[0] = 1;
}
__global__ void k_barrier(int* p) { __syncthreads(); if (p && threadIdx.x == 12345) p[0] = 1; }

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 8.0;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int wgs = argc > 3 ? atoi(argv[3]) : 30000, thr = argc > 4 ? atoi(argv[4]) : 256;
    int* d; hipMalloc(&d, 64);
    printf("running\n"); fflush(stdout);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int i = 0; i < 50; ++i) {
            if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(thr), 0, 0, d);
            else if (mode == 1) hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(thr), 0, 0, d);
            else if (mode == 2) hipLaunchKernelGGL(k_sleep, dim3(wgs), dim3(thr), 0, 0, d);
            else hipLaunchKernelGGL(k_barrier, dim3(wgs), dim3(thr), 0, 0, d);
        }
        hipDeviceSynchronize(); launches += 50;
    }
    printf("mode %d: %ld launches of %d x %d threads in %.1f s\n", mode, launches, wgs, thr, secs);
    return 0;
}
