// Shared host-side plumbing of libvitadapter_hip.so: error reporting and launch timing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitadapter_hip.h"

namespace vah {

// Stores a printf-formatted message in the calling thread's error slot and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// RAII launch timer: when profiling is enabled it records an event pair around the
// kernel launch(es) issued while it is alive, on `stream`.
class LaunchScope {
  public:
    LaunchScope(const char *name, int64_t algorithmic_bytes, hipStream_t stream);
    ~LaunchScope();

  private:
    int slot_;
    hipStream_t stream_;
};

// Checks hipGetLastError() after a launch; returns 0 or the hipError_t (message stored).
int check_launch(const char *what);

constexpr int kCUs = 256;   // MI355X: 8 XCDs x 32 CUs
constexpr int kWave = 64;

}  // namespace vah
