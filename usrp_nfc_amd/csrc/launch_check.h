// launch_check.h -- every kernel launch of the library is followed by hipGetLastError(): a launch the runtime rejects (a
// dynamic-LDS request beyond what the kernel may use, a bad grid, the wrong architecture) leaves no trace in the stream -- the
// next synchronisation succeeds and the host would read the PREVIOUS batch's totals and verdict out of its mirror.  The first
// failure is kept per host thread until the batch that made it reports it (nfc_amd.hip: launch_failed -> NFC_ERR_DEVICE).
#pragma once
#include <hip/hip_runtime.h>

namespace nfc {

struct LaunchError {
    hipError_t err = hipSuccess;
    const char *file = "";
    int line = 0;
};
inline LaunchError &launch_error() {
    static thread_local LaunchError e;
    return e;
}
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
