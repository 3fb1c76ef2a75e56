// status.h -- error plumbing shared by the translation units of libvlfm_amd.so
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/vlfm_amd.h"

namespace vlfm {
void set_last_error(const char* msg);
inline int fail(int code, const char* msg) {
    set_last_error(msg);
    return code;
}
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(hipGetErrorString(e));
        (void)what;
        return VLFM_ERR_HIP;
    }
    return VLFM_OK;
}
}  // namespace vlfm
