// ct_common.hip -- version, error strings, device info for the C ABI (include/ctamd.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

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

}  // extern "C"
