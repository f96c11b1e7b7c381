// Cross-file hooks of the MSDA kernels (not part of the C ABI).
#pragma once
#include "common.h"

namespace vah {

// d(loc), d(attn) of the plain fp32 backward WITHOUT the grad_value scatter (msda.hip: msda_bwd_lanec<32, PU, false>;
// spec cuh:301-403 minus its col2im atomics).  D must be 32.
int msda_grad_taps_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attn,
                       const float *grad_out, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                       float *grad_loc, float *grad_attn, hipStream_t st);

// d(offsets), d(logits) of the fused core, nothing scattered (msda_fused.hip: msda_fused_bwd_vec4 / msda_fused_bwd).
int msda_fused_grad_taps(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi, const void *offsets,
                         const void *logits, int param_dtype, const float *ref, int64_t ref_levels, const void *grad_out,
                         int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq, int64_t P, void *d_offsets, void *d_logits,
                         hipStream_t st);

}  // namespace vah
