// ct_fill.h -- stream-ordered fills by a kernel of this library (every translation unit gets its own copy).
//
// Why not hipMemsetAsync: a hipMemsetAsync of a few bytes (a 128-byte flag block cleared between two kernels of one stream) was observed to
// take effect OUT OF ORDER with its neighbours while another stream kept the GPU busy -- the accurate correction then saw the previous
// round's flags and ran a round more or less (scripts/probe/corr_beside_unet.py: 80 of 80 calls beside a stream of torch fill_ kernels,
// 10-30 % beside the U-Net, never on an idle GPU; gone when the same words are cleared by a kernel).  Kernels of one stream do run in
// order, so every clear the results depend on is a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__global__ __launch_bounds__(256) void ct_fill_kernel(unsigned char* __restrict__ p, size_t bytes, uint32_t word) {
    // head bytes up to 16-byte alignment, 16-byte body, tail bytes: each part by the threads that own it
    const size_t head = (16 - ((uintptr_t)p & 15)) & 15;
    const size_t h = head < bytes ? head : bytes;
    const size_t nvec = (bytes - h) / 16;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    uint4* body = reinterpret_cast<uint4*>(p + h);
    for (size_t i = t; i < nvec; i += nt) body[i] = uint4{word, word, word, word};
    const size_t done = h + nvec * 16;
    if (t < h) p[t] = (unsigned char)word;
    if (t < bytes - done) p[done + t] = (unsigned char)word;
}

// memset(p, byte_value, bytes) on `stream`
inline hipError_t ct_fill_async(void* p, int byte_value, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    const uint32_t b = (uint32_t)(byte_value & 0xFF), word = b | (b << 8) | (b << 16) | (b << 24);
    size_t blocks = (bytes / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ct_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char*)p, bytes, word);
    return hipGetLastError();
}

}  // namespace
