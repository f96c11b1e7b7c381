/*
 * oracle/msda_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's multi-scale
 * deformable attention arithmetic.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path (the HIP library
 * under vit-adapter_amd/csrc) never links or calls it.
 *
 * Reference behaviour restated here (paths under /root/reference/detection/ops/src/cuda):
 *   forward  : ms_deform_im2col_cuda.cuh:237-299 (per-output loop) and :33-84 (bilinear read)
 *   backward : ms_deform_im2col_cuda.cuh:301-403 (per (l,p) reduction over channels) and
 *              :87-159 (4-corner scatter + loc / weight partials)
 *
 * Semantics kept exactly:
 *   - pixel coordinates  h_im = loc_y*H - 0.5 , w_im = loc_x*W - 0.5   (cuh:285-286)
 *   - a sample contributes only if -1 < h_im < H and -1 < w_im < W     (cuh:288)
 *   - each of the four corners is bounds-checked on its own; an out-of-range
 *     corner contributes zero (cuh:55-78)
 *   - grad_loc[...,0] = W * sum_c(gw_c) * g_c * w ; grad_loc[...,1] = H * ...   (cuh:157-158)
 *   - grad_attn_weight = sum_c g_c * bilinear_c                                 (cuh:156)
 * The accumulation order over channels / samples is sequential here (the GPU
 * reference is order-nondeterministic for grad_value because of atomics), so
 * comparisons against GPU results are tolerance-based, never bit-based.
 *
 * Pinned by tests/test_oracle_golden.py against tests/golden/msda_*.npz, which were
 * produced by the reference's own ms_deform_attn_core_pytorch (+ autograd) through
 * tools/gen_golden.py.
 *
 * The file is compiled twice through the REAL macro below: once for float, once
 * for double.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define SUFFIX _f32
#include "msda_oracle_body.inc"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX _f64
#include "msda_oracle_body.inc"
#undef REAL
#undef SUFFIX
