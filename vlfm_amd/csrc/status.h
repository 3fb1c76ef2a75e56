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
// Compute units of the CURRENT device, asked once per device and process (launch-size decisions: "one workgroup per CU").
inline int device_cu_count() {
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;   // MI355X
        }
        cached[dev] = n;
    }
    return cached[dev];
}
// More than 64 KB of dynamic LDS needs an explicit opt-in per kernel AND per device.  One instance per call site
// (function-local static); remembers how many bytes each device of this process has been opted in to, and asks again only
// for more.  (The limit is 160 KB minus the kernel's static LDS: ask for what the launch uses, not for the maximum.)
struct LdsOptIn {
    size_t have[64] = {};
    bool ensure(const void* kernel, size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (dev >= 0 && dev < 64 && have[dev] >= bytes) return true;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (dev >= 0 && dev < 64) have[dev] = bytes;
        return true;
    }
};
}  // namespace vlfm
