// ct_fill.h -- stream-ordered fills by a kernel of this library (every translation unit gets its own copy).
//
// Used where a few flag words are cleared between two kernels of one stream and the result depends on it (csrc/ct_correct.hip).  While
// the correction's irreproducibility beside other streams was being hunted, one box showed it in 80 of 80 calls with
// hipMemsetAsync clearing the flag block and in none with this kernel; other boxes did not repeat that, and the cause that was finally
// pinned down lies elsewhere (stale wave-uniform loads).  The clears stay kernels of the library: same cost, one unknown less.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
    static const size_t cap = [] { const char* e = getenv("CT_FILL_BLOCKS"); const long v = e ? atol(e) : 4096; return (size_t)(v >= 1 ? v : 4096); }();   // (grid-stride: any cap works)
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ct_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char*)p, bytes, word);
    return hipGetLastError();
}

}  // namespace
