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
// More than 64 KB of dynamic LDS needs an explicit opt-in per kernel AND per device.  One instance per call site
// (function-local static); remembers which devices of this process have been done.
struct LdsOptIn {
    unsigned long long done = 0;
    bool ensure(const void* kernel, size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (dev < 64 && ((done >> dev) & 1ull)) return true;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
        if (dev < 64) done |= 1ull << dev;
        return true;
    }
};
}  // namespace vlfm
