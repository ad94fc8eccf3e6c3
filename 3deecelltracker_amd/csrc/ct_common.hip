// ct_common.hip -- version, error strings, device info for the C ABI (include/ctamd.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdint.h>

#include "../../include/ctamd.h"

extern "C" {

int ct_version(void) { return 100; }

const char* ct_error_string(int code) {
    switch (code) {
        case CT_OK: return "ok";
        case CT_EINVAL: return "invalid argument";
        case CT_ESHAPE: return "unsupported shape";
        case CT_EWORKSPACE: return "workspace too small";
        case CT_ENOTCONV: return "iteration bound reached";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

int ct_device_info(int device, int* n_cu, size_t* hbm_bytes, char* name, size_t name_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return (int)e;
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (name && name_len) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    return CT_OK;
}


// Streams restricted to a contiguous range of CUs [first_cu, first_cu + n_cu).  The latency-bound matching
// chain (hundreds of tiny dependent kernels) and the throughput-bound U-Net share one GPU: giving each its
// own CUs lets the small kernels start immediately instead of queueing behind resident conv workgroups.
int ct_stream_create_cu_range(int device, int first_cu, int n_cu, ct_stream_t* out) {
    if (!out || first_cu < 0 || n_cu <= 0) return CT_EINVAL;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return (int)e;
    const int total = prop.multiProcessorCount;
    if (first_cu + n_cu > total) return CT_EINVAL;
    const int words = (total + 31) / 32;
    uint32_t mask[32] = {0};
    if (words > 32) return CT_ESHAPE;
    for (int c = first_cu; c < first_cu + n_cu; ++c) mask[c / 32] |= (1u << (c % 32));
    hipStream_t st;
    e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
    if (e != hipSuccess) return (int)e;
    *out = (ct_stream_t)st;
    return CT_OK;
}

int ct_stream_destroy(ct_stream_t s) { return s ? (int)hipStreamDestroy((hipStream_t)s) : CT_EINVAL; }

}  // extern "C"
