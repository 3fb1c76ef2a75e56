// profile.h -- optional per-kernel HIP-event timing inside the library (bench.py's roofline.achieved).
// When enabled, every kernel launch goes through hipExtLaunchKernelGGL with a start and a stop hipEvent attached to the
// DISPATCH itself, on the launch stream: the events carry the kernel's own begin/end timestamps (the same clock
// rocprofv3's kernel trace reports), with no extra marker packets in front of or behind the kernel.  Disabled (the
// default) it costs one thread-local read per launch.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace vlfm {
struct ProfileScope {
    ProfileScope(const char* name, hipStream_t stream);
    ~ProfileScope();
    const char* name_;
    hipStream_t stream_;
    hipEvent_t start_ = nullptr, stop_ = nullptr;
    bool active_ = false;    // events exist; the next VLFM_KLAUNCH in this scope consumes them
    bool used_ = false;
    ProfileScope* prev_ = nullptr;
};
ProfileScope* current_profile_scope();
}  // namespace vlfm
#define VLFM_TIMED(name, stream) vlfm::ProfileScope _vlfm_scope_##__LINE__(name, (hipStream_t)(stream))
// Launch `kernel`; inside an active VLFM_TIMED scope the first launch is the timed one.
#define VLFM_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                                        \
    do {                                                                                                             \
        vlfm::ProfileScope* _ps = vlfm::current_profile_scope();                                                     \
        if (_ps && _ps->active_ && !_ps->used_) {                                                                    \
            _ps->used_ = true;                                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), _ps->start_, _ps->stop_, 0,      \
                                  __VA_ARGS__);                                                                      \
        } else {                                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__);                      \
        }                                                                                                            \
    } while (0)
