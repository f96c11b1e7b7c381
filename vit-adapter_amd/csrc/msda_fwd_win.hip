// Fused MSDeformAttn forward over LDS value windows, single-level calls (the extractor of ViT-Adapter:
// 21 504 queries of three query grids sampling ONE 64 x 64 value map, /root/reference/segmentation/
// mmseg_custom/models/backbones/adapter_modules.py:28-47, 138-152).
//
// Same arithmetic as msda_fused_fwd (softmax over the P logits, loc = ref + off / (W, H), bilinear gather:
// spec ms_deform_im2col_cuda.cuh:33-84, 237-299); what changes is who does what:
//   * the queries are grouped by the 8 x 8-pixel tile of the value map their reference point falls in (host
//     schedule, static per reference grid: `perm`, `group_off`, `gwin`); a workgroup = (batch n, group, head m)
//     stages the group's value window - the tile + `halo` + 1 pixels on every side, 64-byte rows of one head -
//     in LDS ONCE (~20 x 20 rows, 25 KB) and every one of the group's ~336 queries reads its 16 corner rows
//     from there: 12x reuse of a row instead of 12 passes through L2;
//   * ONE LANE PER (query, head) ROW, not 8: the tap arithmetic, the weights and the address of a corner are
//     computed once instead of on 8 lanes, and the 32 channel sums stay in the lane's registers (no cross-lane
//     reduction).  The 8-lane kernel spends ~20 wave-instructions per sample, this one ~3.
// Corners outside the window (offsets beyond `halo`) are read from global memory by the lane that needs them,
// so any offsets give the same result; they only cost time.
#include <type_traits>

#include "msda_common.h"

namespace vah {
namespace {

using namespace vah::msda;

constexpr int kD = 32;
constexpr int kP = 4;
constexpr int kWinThreads = 128;

typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <typename T>
__device__ __forceinline__ float word_elem(const uint32_t *w, int i) {       // element i of a packed T array
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(float, w[i]);
    else return __builtin_bit_cast(float, (i & 1) ? (w[i >> 1] & 0xFFFF0000u) : (w[i >> 1] << 16));
}

// acc[0..31] += w * row, row = 32 channels of VT at p (LDS or global, 16-byte aligned)
template <typename VT>
__device__ __forceinline__ void axpy_row(float (&acc)[kD], float w, const unsigned char *p) {
    constexpr int PCS = kD * (int)sizeof(VT) / 16;
    uint4 v[PCS];
#pragma unroll
    for (int i = 0; i < PCS; ++i) v[i] = *reinterpret_cast<const uint4 *>(p + 16 * i);
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
        const uint32_t wd[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        constexpr int E = 16 / (int)sizeof(VT);                  // elements per piece
#pragma unroll
        for (int e = 0; e < E; ++e) acc[i * E + e] += w * word_elem<VT>(wd, e);
    }
}

struct GroupWin {       // per group: window origin and size in pixels of the level
    int y0, x0, h, w;
};

template <typename VT, typename PT>
__global__ __launch_bounds__(kWinThreads) void msda_fused_fwd_win(
    const VT *__restrict__ value, const PT *__restrict__ off, const PT *__restrict__ logit, const float *__restrict__ ref,
    const int *__restrict__ perm, const int *__restrict__ group_off, const GroupWin *__restrict__ gwin, int H, int W,
    int64_t start, int64_t S, int M, int64_t Lq, int ngroups, int64_t nblocks, VT *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char Vs[];
    constexpr int ROWB = kD * (int)sizeof(VT);
    constexpr int PCS = ROWB / 16;
    const int64_t blk = xcd_chunked_block(nblocks);
    if (blk >= nblocks) return;
    const int m = (int)(blk % M);
    const int g = (int)((blk / M) % ngroups);
    const int64_t n = blk / M / ngroups;
    const GroupWin gw = gwin[g];
    const int64_t stride = (int64_t)M * kD;
    const VT *vmap = value + (n * S + start) * stride + m * kD;
    // ---- the window: whole 64 / 128-byte rows, 16 bytes per lane and load
    for (int i = threadIdx.x; i < gw.h * gw.w * PCS; i += kWinThreads) {
        const int wp = i / PCS, pc = i - wp * PCS;
        const int wy = wp / gw.w, wx = wp - wy * gw.w;
        *reinterpret_cast<uint4 *>(Vs + wp * ROWB + 16 * pc) = *reinterpret_cast<const uint4 *>(
            reinterpret_cast<const unsigned char *>(vmap + ((int64_t)(gw.y0 + wy) * W + gw.x0 + wx) * stride) + 16 * pc);
    }
    __syncthreads();
    const int beg = group_off[g], end = group_off[g + 1];
    for (int idx = beg + threadIdx.x; idx < end; idx += kWinThreads) {
        const int64_t q = perm[idx];
        const int64_t row = (n * Lq + q) * M + m;
        // ---- the row's operands: 4 offsets (x, y), 4 logits, the reference point
        uint32_t ow[kP * 2 * sizeof(PT) / 4], lw[kP * sizeof(PT) / 4];
        {
            const uint4 *op = reinterpret_cast<const uint4 *>(off + row * kP * 2);
#pragma unroll
            for (int i = 0; i < (int)(kP * 2 * sizeof(PT) / 16); ++i) {
                const uint4 v = op[i];
                ow[4 * i] = v.x, ow[4 * i + 1] = v.y, ow[4 * i + 2] = v.z, ow[4 * i + 3] = v.w;
            }
            const uint2 *lp = reinterpret_cast<const uint2 *>(logit + row * kP);
#pragma unroll
            for (int i = 0; i < (int)(kP * sizeof(PT) / 8); ++i) {
                const uint2 v = lp[i];
                lw[2 * i] = v.x, lw[2 * i + 1] = v.y;
            }
        }
        const float2 rp = *reinterpret_cast<const float2 *>(ref + q * 2);
        float a[kP];
        {
            float mx = -INFINITY, sum = 0.f;
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                a[p] = word_elem<PT>(lw, p);
                mx = fmaxf(mx, a[p]);
            }
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                a[p] = __expf(a[p] - mx);
                sum += a[p];
            }
            const float inv = 1.f / sum;
#pragma unroll
            for (int p = 0; p < kP; ++p) a[p] *= inv;
        }
        float acc[kD];
#pragma unroll
        for (int c = 0; c < kD; ++c) acc[c] = 0.f;
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            // the same expressions as msda_fused_fwd: ref + off / W, then make_tap's arithmetic
            const float lx = rp.x + word_elem<PT>(ow, 2 * p) / (float)W, ly = rp.y + word_elem<PT>(ow, 2 * p + 1) / (float)H;
            const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;
            const float hf = floorf(hs), wf = floorf(ws);
            const int y0 = (int)hf, x0 = (int)wf;
            const float lh = hs - hf, lwt = ws - wf, hh = 1.f - lh, hw = 1.f - lwt;
            const float cw[4] = {hh * hw, hh * lwt, lh * hw, lh * lwt};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
                const bool valid = inside && yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1;
                const int wy = yy - gw.y0, wx = xx - gw.x0;
                const bool inwin = valid && (unsigned)wy < (unsigned)gw.h && (unsigned)wx < (unsigned)gw.w;
                const float wgt = a[p] * cw[c];
                axpy_row<VT>(acc, inwin ? wgt : 0.f, Vs + (inwin ? wy * gw.w + wx : 0) * ROWB);
                if (valid && !inwin)           // beyond the halo: this lane fetches the row itself
                    axpy_row<VT>(acc, wgt, reinterpret_cast<const unsigned char *>(vmap + ((int64_t)yy * W + xx) * stride));
            }
        }
        VT *dst = out + row * kD;
        if constexpr (std::is_same<VT, float>::value) {
#pragma unroll
            for (int c = 0; c < kD; c += 4) *reinterpret_cast<float4 *>(dst + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
        } else {
#pragma unroll
            for (int c = 0; c < kD; c += 8) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[c + e];
                *reinterpret_cast<bf16x8 *>(dst + c) = o;
            }
        }
    }
}

template <typename VT, typename PT>
int launch(const void *value, const void *off, const void *logit, const float *ref, const int *perm, const int *group_off,
           const int *gwin, int H, int W, int64_t start, int64_t N, int64_t S, int64_t M, int64_t Lq, int ngroups, int max_win_px,
           void *out, hipStream_t st) {
    const int64_t nblocks = N * ngroups * M;
    const int64_t grid = (nblocks + 7) / 8 * 8;
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda fused forward (windows): grid too large");
    const int smem = max_win_px * kD * (int)sizeof(VT);
    if (smem > 64 * 1024) return fail(VAH_E_SHAPE, "msda fused forward (windows): window of %d pixels too large", max_win_px);
    if (int rc = allow_dynamic_lds((const void *)msda_fused_fwd_win<VT, PT>, smem, "msda fused forward (windows)")) return rc;
    hipLaunchKernelGGL((msda_fused_fwd_win<VT, PT>), dim3((unsigned)grid), dim3(kWinThreads), smem, st, (const VT *)value,
                       (const PT *)off, (const PT *)logit, ref, perm, group_off, (const GroupWin *)gwin, H, W, start, S, (int)M, Lq,
                       ngroups, nblocks, (VT *)out);
    return check_launch("msda fused forward (windows) launch");
}

}  // namespace
}  // namespace vah

extern "C" {

int vah_msda_fused_forward_win(const void *value, int value_dtype, const void *offsets, const void *logits, int param_dtype,
                               const float *ref, const int32_t *perm, const int32_t *group_off, const int32_t *group_win,
                               int64_t ngroups, int64_t max_win_px, int64_t H, int64_t W, int64_t level_start, int64_t N,
                               int64_t S, int64_t M, int64_t D, int64_t Lq, int64_t P, void *out, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_fused_forward_win";
    if (N < 0 || S < 1 || M < 1 || Lq < 0 || ngroups < 1 || H < 1 || W < 1 || level_start < 0 || level_start + H * W > S ||
        max_win_px < 1 || M * D >= (1LL << 31) || Lq >= (1LL << 31))
        return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (D != kD || P != kP) return fail(VAH_E_UNSUPPORTED, "%s: needs D == 32, P == 4 (one level)", fn);
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !offsets || !logits || !ref || !perm || !group_off || !group_win || !out) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)value | (uintptr_t)out | (uintptr_t)offsets) % 16 || ((uintptr_t)logits | (uintptr_t)ref) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    hipStream_t st = (hipStream_t)stream;
    const int64_t vs = value_dtype == 1 ? 2 : 4, ps = param_dtype == 1 ? 2 : 4;
    LaunchScope scope("msda_fused_fwd", vs * (N * S * M * D + N * Lq * M * D) + ps * 3 * N * Lq * M * P, st,
                      4 * (N * S * M * D + 3 * N * Lq * M * P + N * Lq * M * D));
#define VAH_CASE(VT, VC, PT, PC)                                                                                     \
    if (value_dtype == VC && param_dtype == PC)                                                                      \
        return launch<VT, PT>(value, offsets, logits, ref, perm, group_off, group_win, (int)H, (int)W, level_start, N, S, M, Lq, \
                              (int)ngroups, (int)max_win_px, out, st)
    VAH_CASE(float, 0, float, 0);
    VAH_CASE(__bf16, 1, __bf16, 1);
    VAH_CASE(__bf16, 1, float, 0);
    VAH_CASE(float, 0, __bf16, 1);
#undef VAH_CASE
    return fail(VAH_E_UNSUPPORTED, "%s: dtype codes must be 0 (f32) or 1 (bf16)", fn);
}

}  // extern "C"
