// Multi-scale deformable attention, "windowed" kernels for gfx950: the value rows (and, in the
// backward, the grad_value rows) that a spatially compact group of queries touches are kept in
// LDS, so the 4-corner gathers / scatters hit LDS instead of L2 / the memory-side atomic units.
//
// Same arithmetic as msda.hip (spec: /root/reference/detection/ops/src/cuda/
// ms_deform_im2col_cuda.cuh:33-159,237-403); results differ from the plain kernels only by fp32
// summation order.
//
// Work decomposition
//   * the host passes a query schedule: `perm` lists the query indices group by group,
//     `group_off[g] .. group_off[g+1]` delimits group g.  Groups are meant to be spatially
//     compact (the Python side builds them from the reference points: all queries whose
//     reference point falls into one tile of the value map).  Correctness never depends on the
//     schedule: ANY permutation / grouping of [0, Lq) gives the same result, it only decides how
//     many samples hit the LDS windows.
//   * one workgroup = (batch n, group g, head m), 1024 threads.
//       A  one lane per SAMPLE: read its location / weight, take the per-level bounding box of
//          the touched pixels (registers -> wave reduce -> 4 LDS atomics per wave)
//       B  clip the boxes to the LDS budget -> window rectangles
//       C  one lane per sample again: turn every sample into a 32-byte DESCRIPTOR in LDS
//          {lh, lw, weight, packed (h_low, w_low), LDS index of each of the 4 corners or a
//          "global" / "invalid" marker}.  All bilinear / bounds / index arithmetic is done here
//          once per sample by one lane, not 32 times by the 32 channel lanes of phase D;
//          meanwhile the value windows are staged (16-byte loads, 8 in flight per lane)
//       D  channel-parallel: 32 lanes (backward) or 8 lanes x float4 (forward) per query walk
//          the descriptors; corners inside a window are served from LDS (ds_read / ds_add_f32),
//          the rare corners outside fall back to global loads / global atomics exactly like the
//          plain kernels
//       E  (backward) the grad windows are flushed with one global_atomic_add_f32 per element,
//          whole 128-byte rows per half-wave instruction, instead of one per sample and corner.
//   * gradients w.r.t. sampling locations / attention weights are reduced over the 32 channel
//     lanes with DPP (VALU) adds: no LDS traffic, no barrier per sample.
#include "common.h"

#include <climits>

namespace vah {
namespace {

constexpr int kD = 32;            // channels per head on this path
constexpr int kMaxLevels = 4;
constexpr int kThreads = 1024;    // 16 waves
constexpr int kInvalid = -1;      // corner outside the map / sample outside the gate
constexpr int kGlobal = -2;       // corner valid but outside the LDS window

struct Win {
    int h0, w0, nh, nw;   // window rectangle in the level's pixel grid (nh*nw may be 0)
    int base;             // offset (in floats) of the window inside the LDS window area
    int H, W;             // level geometry
    int valid;            // level passes the guard
    long long start;      // level start row
};

// 32 bytes per sample, written in phases A/C, read (broadcast) in phase D
struct __attribute__((aligned(16))) Desc {
    float lh, lw, a;
    int hw;               // (h_low + 1) << 16 | (w_low + 1); only used by the global fallback
    int idx[4];           // LDS float index of the corner's row, or kInvalid / kGlobal
};

__device__ __forceinline__ void lds_add(float *p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Tap {
    int h_low, w_low;
    float lh, lw;
    bool inside;
};

// Pixel coordinates of a sample and its gate, exactly as make_tap in msda.hip.
__device__ __forceinline__ Tap make_tap(float lx, float ly, int H, int W) {
    Tap t;
    const float h_im = ly * (float)H - 0.5f;
    const float w_im = lx * (float)W - 0.5f;
    t.inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hs = t.inside ? h_im : 0.f;
    const float ws = t.inside ? w_im : 0.f;
    const float hf = floorf(hs), wf = floorf(ws);
    t.h_low = (int)hf;
    t.w_low = (int)wf;
    t.lh = hs - hf;
    t.lw = ws - wf;
    return t;
}

// LDS carve-up of the dynamic area: [windows: copies * budget_px * 32 floats][descriptors]
struct Carve {
    float *win;
    Desc *desc;
};
__device__ __forceinline__ Carve carve(float *smem, int budget_px, int copies) {
    Carve c;
    c.win = smem;
    c.desc = reinterpret_cast<Desc *>(smem + (size_t)budget_px * kD * copies);
    return c;
}

// Phases A-C.  Leaves the descriptors of the group's nq*LP samples in `desc` and (if STAGE) the
// value windows in `s_val`.  Ends with a barrier.
template <bool STAGE>
__device__ void prepare(Win *s_win, int *s_bb, Desc *desc, float *s_val,
                        const float *__restrict__ value, const long long *__restrict__ shapes,
                        const long long *__restrict__ lsi, const float *__restrict__ loc,
                        const float *__restrict__ attn, const int *__restrict__ perm, int goff,
                        int nq, long long n, long long S, int M, int m, int L, long long Lq, int P,
                        int budget_px) {
    const int LP = L * P;
    const int nitems = nq * LP;
    if (threadIdx.x < kMaxLevels * 4) s_bb[threadIdx.x] = (threadIdx.x & 1) ? INT_MIN : INT_MAX;
    if (threadIdx.x < L) {
        const int l = threadIdx.x;
        const long long H = shapes[2 * l], W = shapes[2 * l + 1], st = lsi[l];
        Win w;
        w.valid = H >= 1 && W >= 1 && st >= 0 && H <= S && W <= S && st + H * W <= S && H < 32767 &&
                  W < 32767;
        w.H = (int)H;
        w.W = (int)W;
        w.start = st;
        w.h0 = w.w0 = w.nh = w.nw = w.base = 0;
        s_win[l] = w;
    }
    __syncthreads();

    // ---- A: one lane per sample: load, gate, bounding boxes ------------------------------
    int bmin_h[kMaxLevels], bmax_h[kMaxLevels], bmin_w[kMaxLevels], bmax_w[kMaxLevels];
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) {
        bmin_h[l] = bmin_w[l] = INT_MAX;
        bmax_h[l] = bmax_w[l] = INT_MIN;
    }
    for (int i = threadIdx.x; i < nitems; i += kThreads) {
        const int qi = i / LP, s = i - qi * LP, l = s / P;
        const long long q = perm[goff + qi];
        Desc d;
        d.lh = d.lw = d.a = 0.f;
        d.hw = 0;
        d.idx[0] = d.idx[1] = d.idx[2] = d.idx[3] = kInvalid;
        const Win w = s_win[l];
        if (q >= 0 && q < Lq && w.valid) {
            const long long row = (n * Lq + q) * M + m;
            const float2 xy = *reinterpret_cast<const float2 *>(loc + (row * LP + s) * 2);
            const Tap t = make_tap(xy.x, xy.y, w.H, w.W);
            if (t.inside) {
                d.a = attn[row * LP + s];
                d.lh = t.lh;
                d.lw = t.lw;
                d.hw = ((t.h_low + 1) << 16) | (t.w_low + 1);
                d.idx[0] = kGlobal;             // "live" marker, resolved in phase C
#pragma unroll
                for (int k = 0; k < kMaxLevels; ++k)
                    if (k == l) {
                        bmin_h[k] = min(bmin_h[k], t.h_low);
                        bmax_h[k] = max(bmax_h[k], t.h_low + 1);
                        bmin_w[k] = min(bmin_w[k], t.w_low);
                        bmax_w[k] = max(bmax_w[k], t.w_low + 1);
                    }
            }
        }
        desc[i] = d;
    }
    for (int l = 0; l < L; ++l) {
        int v0 = bmin_h[l], v1 = bmax_h[l], v2 = bmin_w[l], v3 = bmax_w[l];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            v0 = min(v0, __shfl_xor(v0, o, 64));
            v1 = max(v1, __shfl_xor(v1, o, 64));
            v2 = min(v2, __shfl_xor(v2, o, 64));
            v3 = max(v3, __shfl_xor(v3, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_bb[l * 4 + 0], v0);
            atomicMax(&s_bb[l * 4 + 1], v1);
            atomicMin(&s_bb[l * 4 + 2], v2);
            atomicMax(&s_bb[l * 4 + 3], v3);
        }
    }
    __syncthreads();

    // ---- B: windows ----------------------------------------------------------------------
    if (threadIdx.x == 0) {
        int nh[kMaxLevels], nw[kMaxLevels], h0[kMaxLevels], w0[kMaxLevels];
        long long total = 0;
        for (int l = 0; l < L; ++l) {
            nh[l] = nw[l] = h0[l] = w0[l] = 0;
            if (!s_win[l].valid || s_bb[l * 4 + 1] < s_bb[l * 4 + 0]) continue;
            const int a = max(s_bb[l * 4 + 0], 0), b = min(s_bb[l * 4 + 1], s_win[l].H - 1);
            const int c = max(s_bb[l * 4 + 2], 0), d = min(s_bb[l * 4 + 3], s_win[l].W - 1);
            if (b < a || d < c) continue;
            h0[l] = a;
            w0[l] = c;
            nh[l] = b - a + 1;
            nw[l] = d - c + 1;
            total += (long long)nh[l] * nw[l];
        }
        // over budget: shrink every window around its centre by the same linear factor until the
        // sum fits (samples that fall outside take the global path; results are unchanged).
        int guard = 0;
        while (total > budget_px && guard++ < 64) {
            const float f = sqrtf((float)budget_px / (float)total) * 0.97f;
            total = 0;
            for (int l = 0; l < L; ++l) {
                if (nh[l] == 0) continue;
                const int nh2 = max(1, (int)((float)nh[l] * f)), nw2 = max(1, (int)((float)nw[l] * f));
                h0[l] += (nh[l] - nh2) / 2;
                w0[l] += (nw[l] - nw2) / 2;
                nh[l] = nh2;
                nw[l] = nw2;
                total += (long long)nh2 * nw2;
            }
            if (total <= (long long)L) break;
        }
        int base = 0;
        for (int l = 0; l < L; ++l) {
            if (total > budget_px) nh[l] = nw[l] = 0;       // pathological: no windows at all
            s_win[l].h0 = h0[l];
            s_win[l].w0 = w0[l];
            s_win[l].nh = nh[l];
            s_win[l].nw = nw[l];
            s_win[l].base = base;
            base += nh[l] * nw[l] * kD;
        }
    }
    __syncthreads();

    // ---- C: resolve corner indices (one lane per sample) ...---------------------------------
    for (int i = threadIdx.x; i < nitems; i += kThreads) {
        Desc d = desc[i];
        if (d.idx[0] != kGlobal) continue;            // dead sample
        const int s = i % LP, l = s / P;
        const Win w = s_win[l];
        const int h_low = (d.hw >> 16) - 1, w_low = (d.hw & 0xFFFF) - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int hk = h_low + (k >> 1), wk = w_low + (k & 1);
            const int rh = hk - w.h0, rw = wk - w.w0;
            int v = kGlobal;
            if (hk < 0 || wk < 0 || hk > w.H - 1 || wk > w.W - 1)
                v = kInvalid;
            else if (rh >= 0 && rh < w.nh && rw >= 0 && rw < w.nw)
                v = w.base + (rh * w.nw + rw) * kD;
            d.idx[k] = v;
        }
        *reinterpret_cast<int4 *>(desc[i].idx) = make_int4(d.idx[0], d.idx[1], d.idx[2], d.idx[3]);
    }

    // ---- ... and stage the value windows: 8 lanes x float4 per pixel row, 8 rows in flight ---
    if (STAGE) {
        constexpr int U = 8;
        const long long stride = (long long)M * kD;
        const int part = threadIdx.x & 7;
        const int lanes_px = kThreads >> 3;
        for (int l = 0; l < L; ++l) {
            const Win w = s_win[l];
            const int npx = w.nh * w.nw;
            const float *vl = value + (n * S + w.start) * stride + m * kD + part * 4;
            for (int j0 = threadIdx.x >> 3; j0 < npx; j0 += lanes_px * U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = min(j0 + u * lanes_px, npx - 1);
                    const int r = j / w.nw, c = j - r * w.nw;
                    v[u] = *reinterpret_cast<const float4 *>(
                        vl + ((long long)(w.h0 + r) * w.W + (w.w0 + c)) * stride);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * lanes_px;
                    if (j < npx) *reinterpret_cast<float4 *>(s_val + w.base + j * kD + part * 4) = v[u];
                }
            }
        }
    }
    __syncthreads();
}

// Global offset (in floats, relative to the level's head pointer) of corner k of a descriptor.
__device__ __forceinline__ long long corner_offset(const Desc &d, int k, int W, long long stride) {
    const int hk = (d.hw >> 16) - 1 + (k >> 1), wk = (d.hw & 0xFFFF) - 1 + (k & 1);
    return ((long long)hk * W + wk) * stride;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void msda_fwd_win(
    const float *__restrict__ value, const long long *__restrict__ shapes,
    const long long *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ attn, const int *__restrict__ group_off,
    const int *__restrict__ perm, int n_groups, int max_group, long long S, int M, int L,
    long long Lq, int P, int budget_px, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ Win s_win[kMaxLevels];
    __shared__ int s_bb[kMaxLevels * 4];
    const int LP = L * P;
    const Carve cv = carve(smem, budget_px, 1);
    float *s_val = cv.win;

    // block -> (n, group, head), head fastest so consecutive blocks share loc/attn cache lines
    const long long b = blockIdx.x;
    const int m = (int)(b % M);
    const int g = (int)((b / M) % n_groups);
    const long long n = b / M / n_groups;
    const int goff = group_off[g], nq = group_off[g + 1] - goff;
    if (nq <= 0 || goff < 0 || (long long)goff + nq > Lq || nq > max_group) return;   // malformed: skip
    const long long stride = (long long)M * kD;

    prepare<true>(s_win, s_bb, cv.desc, s_val, value, shapes, lsi, loc, attn, perm, goff, nq, n, S, M,
                  m, L, Lq, P, budget_px);

    // phase D: 8 lanes x float4 per query row; 128 query rows in flight per pass
    const int sub = threadIdx.x & 7;
    for (int qi = threadIdx.x >> 3; qi < nq; qi += kThreads >> 3) {
        const long long q = perm[goff + qi];
        if (q < 0 || q >= Lq) continue;
        const long long row = (n * Lq + q) * M + m;
        const Desc *dp = cv.desc + qi * LP;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < L; ++l) {
            const Win w = s_win[l];
            const float *vl = value + (n * S + w.start) * stride + m * kD + sub * 4;
            for (int p = 0; p < P; ++p) {
                const Desc d = dp[l * P + p];
                const float hh = 1.f - d.lh, hw = 1.f - d.lw;
                const float cw[4] = {hh * hw * d.a, hh * d.lw * d.a, d.lh * hw * d.a, d.lh * d.lw * d.a};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float4 v;
                    if (d.idx[k] >= 0)
                        v = *reinterpret_cast<const float4 *>(s_val + d.idx[k] + sub * 4);
                    else if (d.idx[k] == kGlobal)
                        v = *reinterpret_cast<const float4 *>(vl + corner_offset(d, k, w.W, stride));
                    else
                        continue;
                    acc.x += cw[k] * v.x;
                    acc.y += cw[k] * v.y;
                    acc.z += cw[k] * v.z;
                    acc.w += cw[k] * v.w;
                }
            }
        }
        *reinterpret_cast<float4 *>(out + row * kD + sub * 4) = acc;
    }
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
template <bool STAGE>
__global__ __launch_bounds__(kThreads) void msda_bwd_win(
    const float *__restrict__ value, const long long *__restrict__ shapes,
    const long long *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ attn, const float *__restrict__ grad_out,
    const int *__restrict__ group_off, const int *__restrict__ perm, int n_groups, int max_group,
    long long S, int M, int L, long long Lq, int P, int budget_px, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ Win s_win[kMaxLevels];
    __shared__ int s_bb[kMaxLevels * 4];
    const int LP = L * P;
    // STAGE: [value windows | grad windows]; otherwise the window area holds grad windows only and
    // value corners are read from global memory (used when both do not fit, e.g. 3-level calls)
    const Carve cv = carve(smem, budget_px, STAGE ? 2 : 1);
    float *s_val = cv.win;
    float *s_grad = STAGE ? cv.win + (size_t)budget_px * kD : cv.win;

    const long long b = blockIdx.x;
    const int m = (int)(b % M);
    const int g = (int)((b / M) % n_groups);
    const long long n = b / M / n_groups;
    const int goff = group_off[g], nq = group_off[g + 1] - goff;
    if (nq <= 0 || goff < 0 || (long long)goff + nq > Lq || nq > max_group) return;   // malformed: skip
    const long long stride = (long long)M * kD;

    // zero the whole grad window area first (its extent is not known before phase B; the area is
    // budget_px rows at most) - overlaps with phase A's global loads
    for (int i = threadIdx.x * 4; i < budget_px * kD; i += kThreads * 4)
        *reinterpret_cast<float4 *>(s_grad + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    prepare<STAGE>(s_win, s_bb, cv.desc, s_val, value, shapes, lsi, loc, attn, perm, goff, nq, n, S, M,
                   m, L, Lq, P, budget_px);

    // phase D: one half-wave (32 channel lanes) per query row
    const int c = threadIdx.x & 31;
    for (int qi = threadIdx.x >> 5; qi < nq; qi += kThreads >> 5) {
        const long long q = perm[goff + qi];
        if (q < 0 || q >= Lq) continue;        // uniform over the 32 channel lanes
        const long long row = (n * Lq + q) * M + m;
        const Desc *dp = cv.desc + qi * LP;
        float *glp = grad_loc + row * LP * 2;
        float *gap = grad_attn + row * LP;
        const float gc = grad_out[row * kD + c];
        for (int l = 0; l < L; ++l) {
            const Win w = s_win[l];
            const float *vl = value + (n * S + w.start) * stride + m * kD + c;
            float *gvl = grad_value + (n * S + w.start) * stride + m * kD + c;
            const float fW = (float)w.W, fH = (float)w.H;
            for (int p = 0; p < P; ++p) {
                const int s = l * P + p;
                const Desc d = dp[s];
                const float hh = 1.f - d.lh, hw = 1.f - d.lw;
                const float cw[4] = {hh * hw, hh * d.lw, d.lh * hw, d.lh * d.lw};
                const float tv = gc * d.a;
                float vv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vv[k] = 0.f;
                    if (d.idx[k] >= 0) {
                        vv[k] = STAGE ? s_val[d.idx[k] + c] : vl[corner_offset(d, k, w.W, stride)];
                        lds_add(s_grad + d.idx[k] + c, cw[k] * tv);
                    } else if (d.idx[k] == kGlobal) {
                        const long long off = corner_offset(d, k, w.W, stride);
                        vv[k] = vl[off];
                        atomicAdd(gvl + off, cw[k] * tv);
                    }
                }
                const float gh = hw * (vv[2] - vv[0]) + d.lw * (vv[3] - vv[1]);
                const float gw = hh * (vv[1] - vv[0]) + d.lh * (vv[3] - vv[2]);
                const float val = cw[0] * vv[0] + cw[1] * vv[1] + cw[2] * vv[2] + cw[3] * vv[3];
                const float pa = dpp_sum32_hi(gc * val);
                const float pw = dpp_sum32_hi(fW * gw * tv);
                const float ph = dpp_sum32_hi(fH * gh * tv);
                if (c == 16) {
                    *reinterpret_cast<float2 *>(glp + 2 * s) = make_float2(pw, ph);
                    gap[s] = pa;
                }
            }
        }
    }
    __syncthreads();

    // phase E: flush the grad windows, one 128-byte row per half-wave instruction
    for (int l = 0; l < L; ++l) {
        const Win w = s_win[l];
        const int npx = w.nh * w.nw;
        float *gvl = grad_value + (n * S + w.start) * stride + m * kD + c;
        for (int j = threadIdx.x >> 5; j < npx; j += kThreads >> 5) {
            const int r = j / w.nw, cc = j - r * w.nw;
            const float v = s_grad[w.base + j * kD + c];
            atomicAdd(gvl + ((long long)(w.h0 + r) * w.W + (w.w0 + cc)) * stride, v);
        }
    }
}

int check_sched(const char *fn, const int *group_off, const int *perm, int64_t n_groups,
                int64_t max_group, int64_t budget_px, int64_t D, int64_t L, int64_t P, int copies,
                size_t *smem) {
    if (!group_off || !perm) return fail(VAH_E_NULL, "%s: null schedule pointer", fn);
    if (n_groups < 1 || n_groups > (1 << 24)) return fail(VAH_E_SHAPE, "%s: bad n_groups", fn);
    if (D != kD) return fail(VAH_E_UNSUPPORTED, "%s: the windowed path needs D == 32", fn);
    if (L > kMaxLevels) return fail(VAH_E_UNSUPPORTED, "%s: the windowed path needs L <= 4", fn);
    if (max_group < 1 || max_group > (1 << 20)) return fail(VAH_E_SHAPE, "%s: bad max_group", fn);
    const int64_t bytes = budget_px * kD * 4 * copies + max_group * L * P * (int64_t)sizeof(Desc);
    if (budget_px < 1 || bytes > 160 * 1024 - 512)
        return fail(VAH_E_SHAPE, "%s: %lld window pixels x%d + %lld queries per group do not fit "
                    "160 KiB of LDS", fn, (long long)budget_px, copies, (long long)max_group);
    *smem = (size_t)bytes;
    return VAH_OK;
}

}  // namespace
}  // namespace vah

extern "C" {

int vah_msda_forward_win_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                             const float *loc, const float *attn, const int32_t *group_off,
                             const int32_t *perm, int64_t n_groups, int64_t max_group,
                             int64_t budget_px, int64_t N, int64_t S, int64_t M, int64_t D,
                             int64_t L, int64_t Lq, int64_t P, float *out, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_forward_win_f32";
    if (N < 0 || S < 1 || M < 1 || L < 1 || Lq < 0 || P < 1 || M * D >= (1LL << 31))
        return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !loc || !attn || !out) return fail(VAH_E_NULL, "%s: null pointer", fn);
    size_t smem = 0;
    if (int rc = check_sched(fn, group_off, perm, n_groups, max_group, budget_px, D, L, P, 1, &smem)) return rc;
    const int64_t grid = N * n_groups * M;
    if (grid >= (1LL << 31)) return fail(VAH_E_SHAPE, "%s: grid too large", fn);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = allow_dynamic_lds((const void *)msda_fwd_win, 160 * 1024 - 512, fn)) return rc;
    const int64_t bytes = 4 * (N * S * M * D + 3 * N * Lq * M * L * P + N * Lq * M * D);
    LaunchScope scope("msda_fwd_f32", bytes, st);
    hipLaunchKernelGGL(msda_fwd_win, dim3((unsigned)grid), dim3(kThreads), smem, st, value,
                       (const long long *)shapes, (const long long *)lsi, loc, attn, group_off, perm,
                       (int)n_groups, (int)max_group, (long long)S, (int)M, (int)L, (long long)Lq,
                       (int)P, (int)budget_px, out);
    return check_launch(fn);
}

int vah_msda_backward_win_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                              const float *loc, const float *attn, const float *grad_out,
                              const int32_t *group_off, const int32_t *perm, int64_t n_groups,
                              int64_t max_group, int64_t budget_px, int stage, int64_t N, int64_t S,
                              int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                              float *grad_value, float *grad_loc, float *grad_attn, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_backward_win_f32";
    if (N < 0 || S < 1 || M < 1 || L < 1 || Lq < 0 || P < 1 || M * D >= (1LL << 31))
        return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    size_t smem = 0;
    if (int rc = check_sched(fn, group_off, perm, n_groups, max_group, budget_px, D, L, P, stage ? 2 : 1, &smem))
        return rc;
    const int64_t grid = N * n_groups * M;
    if (grid >= (1LL << 31)) return fail(VAH_E_SHAPE, "%s: grid too large", fn);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = allow_dynamic_lds(stage ? (const void *)msda_bwd_win<true> : (const void *)msda_bwd_win<false>,
                                   160 * 1024 - 512, fn))
        return rc;
    const int64_t bytes = 4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D);
    LaunchScope scope("msda_bwd_f32", bytes, st);
    if (stage)
        hipLaunchKernelGGL(msda_bwd_win<true>, dim3((unsigned)grid), dim3(kThreads), smem, st, value,
                           (const long long *)shapes, (const long long *)lsi, loc, attn, grad_out,
                           group_off, perm, (int)n_groups, (int)max_group, (long long)S, (int)M, (int)L,
                           (long long)Lq, (int)P, (int)budget_px, grad_value, grad_loc, grad_attn);
    else
        hipLaunchKernelGGL(msda_bwd_win<false>, dim3((unsigned)grid), dim3(kThreads), smem, st, value,
                           (const long long *)shapes, (const long long *)lsi, loc, attn, grad_out,
                           group_off, perm, (int)n_groups, (int)max_group, (long long)S, (int)M, (int)L,
                           (long long)Lq, (int)P, (int)budget_px, grad_value, grad_loc, grad_attn);
    return check_launch(fn);
}

}  // extern "C"
