#include <hip/hip_runtime.h>
__device__ __forceinline__ float wave_max_nonneg(float m) {
    // non-negative floats compare like their bit patterns
    int v = __builtin_bit_cast(int, m);
#define DPPMAX(ctrl, rmask) { int t = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false); v = v > t ? v : t; }
    DPPMAX(0xB1, 0xf)    // quad_perm [1,0,3,2]
    DPPMAX(0x4E, 0xf)    // quad_perm [2,3,0,1]
    DPPMAX(0x141, 0xf)   // row_half_mirror
    DPPMAX(0x140, 0xf)   // row_mirror
    DPPMAX(0x142, 0xa)   // row_bcast:15 -> rows 1, 3
    DPPMAX(0x143, 0xc)   // row_bcast:31 -> rows 2, 3
#undef DPPMAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 63));
}
__global__ void k(const float* in, float* out, float* out2, float* out3) {
    float x = in[threadIdx.x];
    out[threadIdx.x] = wave_max_nonneg(x);
    float a = x, b = x;
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out2[threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    auto q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out3[threadIdx.x] = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
}
int main() {
    float *d, *o, *o2, *o3; hipMalloc(&d, 256*4); hipMalloc(&o, 256*4); hipMalloc(&o2, 256*4); hipMalloc(&o3, 256*4);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37) % 101) + (i >> 6) * 1000;
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, o2, o3);
    float r[256], r2[256], r3[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 1024, hipMemcpyDeviceToHost); hipMemcpy(r3, o3, 1024, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0, bad3 = 0;
    for (int w = 0; w < 4; ++w) { float m = 0; for (int i = 0; i < 64; ++i) m = fmaxf(m, h[w*64+i]);
        for (int i = 0; i < 64; ++i) { if (r[w*64+i] != m) ++bad; if (r2[w*64+i] != h[w*64+i] + h[w*64+(i^32)]) ++bad2; if (r3[w*64+i] != h[w*64+i] + h[w*64+(i^16)]) ++bad3; } }
    printf("dpp max bad %d, permlane32 xor bad %d, permlane16 xor bad %d\n", bad, bad2, bad3);
    return 0;
}
