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
    // algorithmic_bytes: what the launch must move for the IO dtypes it actually runs with;
    // def_bytes: the same operator by its fp32 definition, 0 = same; flops: for MFMA-bound kernels.
    LaunchScope(const char *name, int64_t algorithmic_bytes, hipStream_t stream, int64_t def_bytes = 0,
                int64_t flops = 0);
    ~LaunchScope();

  private:
    int slot_;
    hipStream_t stream_;
};

// Checks hipGetLastError() after a launch; returns 0 or the hipError_t (message stored).
int check_launch(const char *what);

// Raises a kernel's dynamic-LDS limit on the CURRENT device.  The attribute is per device (and cheap to
// set), so it is set before every launch that needs it rather than once per process.
int allow_dynamic_lds(const void *kernel, int bytes, const char *what);

constexpr int kCUs = 256;   // MI355X: 8 XCDs x 32 CUs
constexpr int kWave = 64;

#if defined(__HIPCC__)
// Cross-lane adds on the VALU (DPP) instead of ds_bpermute: __shfl_xor lowers to an LDS-pipeline
// instruction on gfx950, which competes with the LDS atomics / reads of the gather kernels.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL,
                                                                 ROW_MASK, 0xF, false));
}

// Sum over each aligned group of 16 lanes; every lane of the group ends up with the sum.
__device__ __forceinline__ float dpp_sum16(float x) {
    x += dpp_mov<0xB1, 0xF>(x);     // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E, 0xF>(x);     // quad_perm [2,3,0,1]
    x += dpp_mov<0x141, 0xF>(x);    // row_half_mirror
    x += dpp_mov<0x140, 0xF>(x);    // row_mirror
    return x;
}

// Sum over each aligned group of 32 lanes.  The result is valid in the UPPER 16 lanes of the
// group (lanes 16..31 and 48..63): row_bcast15 hands row 0's total to row 1 and row 2's to row 3.
__device__ __forceinline__ float dpp_sum32_hi(float x) {
    x = dpp_sum16(x);
    return x + dpp_mov<0x142, 0xA>(x);
}

// Sum over the 64 lanes of the wave, valid in every lane.
__device__ __forceinline__ float wave_sum(float x) {
    x = dpp_sum16(x);
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
#endif

}  // namespace vah
