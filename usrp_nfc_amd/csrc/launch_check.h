// launch_check.h -- every kernel launch of the library is followed by hipGetLastError(): a launch the runtime rejects (a
// dynamic-LDS request beyond what the kernel may use, a bad grid, the wrong architecture) leaves no trace in the stream -- the
// next synchronisation succeeds and the host would read the PREVIOUS batch's totals and verdict out of its mirror.  The first
// failure is kept in the CONTEXT until the batch that made it reports it (host_context.h: batch_ok -> NFC_ERR_DEVICE).
#pragma once
#include <hip/hip_runtime.h>

namespace nfc {

struct LaunchError {
    hipError_t err = hipSuccess;
    const char *file = "";
    int line = 0;
};
// Where a rejected launch is noted: the context the calling thread is working for (LaunchScope, set by every C-ABI entry that
// takes a context) -- so a batch submitted on one thread and waited for on another still reports it --, else a per-thread slot
// (launches outside any context: the transmit-side renderer).
inline LaunchError *&launch_slot() {
    static thread_local LaunchError *p = nullptr;
    return p;
}
inline LaunchError &launch_error() {
    static thread_local LaunchError fallback;
    return launch_slot() ? *launch_slot() : fallback;
}
struct LaunchScope {
    LaunchError *prev;
    explicit LaunchScope(LaunchError *e) : prev(launch_slot()) { launch_slot() = e; }
    ~LaunchScope() { launch_slot() = prev; }
    LaunchScope(const LaunchScope &) = delete;
    LaunchScope &operator=(const LaunchScope &) = delete;
};
inline void note_launch(const char *file, int line) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && launch_error().err == hipSuccess) launch_error() = LaunchError{e, file, line};
}

}  // namespace nfc

#define NFC_LAUNCH(...)                        \
    do {                                       \
        hipLaunchKernelGGL(__VA_ARGS__);       \
        ::nfc::note_launch(__FILE__, __LINE__); \
    } while (0)
#define NFC_LAUNCH_EXT(...)                    \
    do {                                       \
        hipExtLaunchKernelGGL(__VA_ARGS__);    \
        ::nfc::note_launch(__FILE__, __LINE__); \
    } while (0)
