// profile.h -- optional per-kernel HIP-event timing inside the library (bench.py's roofline.achieved).
// When enabled, every kernel launch is bracketed by two hipEvents recorded on the launch stream, immediately
// before and after the launch, so the measured span is the kernel's own execution (host-side gaps between
// separate API calls are excluded).  Disabled (the default) it costs one branch per launch.
#pragma once
#include <hip/hip_runtime.h>

namespace vlfm {
struct ProfileScope {
    ProfileScope(const char* name, hipStream_t stream);
    ~ProfileScope();
    const char* name_;
    hipStream_t stream_;
    hipEvent_t start_ = nullptr, stop_ = nullptr;
    bool active_ = false;
};
}  // namespace vlfm
#define VLFM_TIMED(name, stream) vlfm::ProfileScope _vlfm_scope_##__LINE__(name, (hipStream_t)(stream))
