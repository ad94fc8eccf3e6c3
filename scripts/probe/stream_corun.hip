// A synthetic bandwidth co-runner for the U-Net (scripts/probe/stream_corun.py): one "sweep" reads a double array and writes another, like the
// watershed's filter passes; mode 0 = plain loads / stores, 1 = non-temporal loads and stores, 2 = non-temporal stores only.
// Question: is what the watershed costs the U-Net (+0.64 ms per frame) its TRAFFIC THROUGH THE CACHES the conv kernels live in (halo sharing and
// the packed weights sit in L2), in which case cache-bypassing sweeps would be cheaper for the same bytes?
#include <hip/hip_runtime.h>
template <int MODE>
__global__ __launch_bounds__(256) void sweep_kernel(const double* __restrict__ in, double* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v;
    if constexpr (MODE == 1) v = __builtin_nontemporal_load(in + i); else v = in[i];
    v = v * 1.0000001 + 0.5;
    if constexpr (MODE >= 1) __builtin_nontemporal_store(v, out + i); else out[i] = v;
}
extern "C" int probe_sweep(const double* in, double* out, long long n, int mode, void* stream) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (mode == 0) hipLaunchKernelGGL(sweep_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, in, out, n);
    else if (mode == 1) hipLaunchKernelGGL(sweep_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, in, out, n);
    else hipLaunchKernelGGL(sweep_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, in, out, n);
    return (int)hipGetLastError();
}
