// Kernel launches of libsoccdpt_hip.so go through SOCCDPT_LAUNCH.  Normally it is hipLaunchKernelGGL.  While a profiling scope is open on
// the calling thread (soccdpt_profile_enable: internal.h ProfScope) every launch instead carries a start / stop event pair bound to the
// DISPATCH ITSELF (hipExtLaunchKernelGGL): hipEventElapsedTime of such a pair is the kernel's own begin -> end time, the figure rocprofv3
// --kernel-trace reports, with no event-record packets between kernels.  (Round 2 bracketed each launch with two hipEventRecord calls;
// every bracket carried ~2 us of dispatch and the per-kernel sums exceeded the un-instrumented step: VERDICT r2 #6.)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace soccdpt {

struct LaunchTimer {
    virtual void next_pair(hipEvent_t* e0, hipEvent_t* e1) = 0;   // a fresh (start, stop) pair for the launch being issued
    virtual ~LaunchTimer() {}
};
LaunchTimer*& launch_timer();   // the calling thread's open profiling scope, or nullptr (capi.cpp)
unsigned long long& launch_counter();   // kernels this thread has launched through SOCCDPT_LAUNCH so far (soccdpt_launch_counter; capi.cpp)

}  // namespace soccdpt

// (two levels: call sites may pass a macro that expands to several arguments)
#define SOCCDPT_LAUNCH(...) SOCCDPT_LAUNCH_I(__VA_ARGS__)
#define SOCCDPT_LAUNCH_I(kernel, grid, block, lds, stream, ...)                                            \
    do {                                                                                                   \
        ++::soccdpt::launch_counter();                                                                     \
        if (::soccdpt::LaunchTimer* _lt = ::soccdpt::launch_timer()) {                                     \
            hipEvent_t _e0 = nullptr, _e1 = nullptr;                                                       \
            _lt->next_pair(&_e0, &_e1);                                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _e0, _e1, 0, __VA_ARGS__);             \
        } else {                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                             \
        }                                                                                                  \
    } while (0)
