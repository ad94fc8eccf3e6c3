// Minimal reproducer attempt for the co-residency hazard recorded in DESIGN.md section 5: waves that execute packed-fp32 VALU
// instructions (victim) beside waves that execute v_mfma_f32_16x16x32_f16 (aggressor) on the same SIMDs, two streams.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/pk_mfma_hazard.hip -o scripts/probe/pk_mfma_hazard && scripts/probe/pk_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void aggressor(float* out, int iters, int mode) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x % 7 + i)); }
    float4v acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc2, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(0.001f * threadIdx.x, 0.002f, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.002f, 0.001f * threadIdx.x, acc2, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc2[2] + acc2[3];
}

// every lane runs the same recurrence on packed pairs; all lanes of a wave must end with the same bits
__global__ __launch_bounds__(256) void victim(float* out, int iters) {
    float2v x = {1.0f, 0.5f}, y = {0.25f, 0.125f};
    const float2v c = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; ++it) {
        x = __builtin_elementwise_fma(x, c, d);          // v_pk_fma_f32
        y = y * c;                                       // v_pk_mul_f32
        x = x + y;                                       // v_pk_add_f32
        y = y + d;
        if (x[0] > 4.f) { x *= 0.25f; y *= 0.25f; }
    }
    out[(blockIdx.x * 256 + threadIdx.x) * 2] = x[0] + y[0];
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = x[1] + y[1];
}

int main() {
    const int VB = 2048, AB = 512;        // 2 aggressor workgroups per CU: the victim must fit beside them
    float *vo, *ao;
    hipMalloc(&vo, VB * 256 * 2 * sizeof(float)); hipMalloc(&ao, AB * 256 * sizeof(float));
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<float> ref(VB * 256 * 2), got(VB * 256 * 2);
    hipLaunchKernelGGL(victim, dim3(VB), dim3(256), 0, s1, vo, 4000); hipStreamSynchronize(s1);
    hipMemcpy(ref.data(), vo, ref.size() * 4, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 2; ++mode) {
        int bad = 0, badlanes[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(aggressor, dim3(AB), dim3(256), 0, s2, ao, 400000, mode);
            hipLaunchKernelGGL(victim, dim3(VB), dim3(256), 0, s1, vo, 4000);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), vo, got.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < got.size(); ++i)
                if (memcmp(&got[i], &ref[i], 4)) { ++bad; ++badlanes[((i / 2) & 63) / 16]; }
        }
        printf("aggressor %s: %d differing values (of %zu x 5); by lane quarter: %d %d %d %d\n", mode == 0 ? "mfma 16x16x32 f16" : "mfma 16x16x4 f32",
               bad, got.size(), badlanes[0], badlanes[1], badlanes[2], badlanes[3]);
    }
    return 0;
}
