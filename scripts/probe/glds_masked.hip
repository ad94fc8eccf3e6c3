#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int n) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 4 * 4; i += 256) lds[i] = -1.f;
    __syncthreads();
    // each wave copies 64 x 16 B; odd lanes masked
    const float* g = src + (size_t)(wave * 64 + lane) * 4;
    float* l = &lds[wave * 64 * 4];
    if ((lane & 1) == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = tid; i < 64 * 4 * 4; i += 256) dst[i] = lds[i];
}
int main() {
    float *s, *d; const int n = 1024;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
    float h[n]; for (int i = 0; i < n; ++i) h[i] = i;
    hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d, n);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) { int lane = (i / 4) & 63; float want = (lane & 1) ? -1.f : (float)i; if (h[i] != want) { if (bad < 5) printf("i %d got %f want %f\n", i, h[i], want); ++bad; } }
    printf("glds masked test: %d mismatches\n", bad);
    return 0;
}
