// Fused MSDeformAttn forward over LDS value windows, single-level calls (the extractor of ViT-Adapter:
// 21 504 queries of three query grids sampling ONE 64 x 64 value map, /root/reference/segmentation/
// mmseg_custom/models/backbones/adapter_modules.py:28-47, 138-152).
//
// Same arithmetic as msda_fused_fwd (softmax over the P logits, loc = ref + off / (W, H), bilinear gather:
// spec ms_deform_im2col_cuda.cuh:33-84, 237-299); what changes is who does what:
//   * the queries are grouped by the 8 x 8-pixel tile of the value map their reference point falls in (schedule built
//     on the device by msda_win_schedule: no host copy of the geometry); a workgroup walks (batch n, group, head m) items,
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

// ---- the schedule, built on the device -------------------------------------------------------------------------
// Workspace: header (16 ints), group_off[G + 1], cursor / count[G], perm[Lq].  Round 2 built perm / group_off / windows
// on the HOST from a D2H copy of spatial_shapes, cached by tensor identity: a sync per forward for callers that make
// fresh shape tensors every call (the reference's deform_inputs does, adapter_modules.py:28-47), and impossible to
// capture in a HIP graph.  Now ONE 1024-thread workgroup counts the queries per 8 x 8-pixel group of the value map
// (LDS counters up to 4096 groups, global ones beyond), scans the counts and writes the permutation; the level geometry
// is read from the device tensors.  The order of the queries inside a group does not matter (every row is computed on
// its own).
struct WinHeader {
    int valid, H, W, ntx, nty, G, pad0, pad1;
    long long start;
    int pad2[6];
};
static_assert(sizeof(WinHeader) == 64, "header");
constexpr int kTile = 8;
constexpr int kSchedThreads = 1024;
constexpr int kLdsGroups = 4096;
constexpr int kRegQ = 32;             // queries a schedule thread keeps in registers (fast path: Lq <= 32 768)

// Group of a reference point: the 8 x 8-pixel tile of the value map it falls in.  Only a heuristic (a corner outside the
// staged window is read from global memory: same result), so one multiply per coordinate - sx = W / 8, sy = H / 8 - and
// no division: the schedule kernel is ONE workgroup and its instruction count is its run time.
__device__ __forceinline__ int group_of(float rx, float ry, float sx, float sy, int ntx, int nty) {
    const int ty = min(max((int)(ry * sy), 0), nty - 1);
    const int tx = min(max((int)(rx * sx), 0), ntx - 1);
    return ty * ntx + tx;
}

__global__ __launch_bounds__(kSchedThreads) void msda_win_schedule(const float *__restrict__ ref, const int64_t *__restrict__ shapes,
                                                                   const int64_t *__restrict__ lsi, int Lq, int64_t S, int Gmax,
                                                                   unsigned char *__restrict__ ws) {
    extern __shared__ int s_dyn[];
    int *s_cnt = s_dyn, *s_scan = s_dyn + kLdsGroups;
    unsigned short *s_g = reinterpret_cast<unsigned short *>(s_scan + kSchedThreads);       // group of query q (fast path)
    // Fast path (the adapter: 21 504 queries, 64 groups; Lq <= 32 768, <= 4096 groups).  This kernel is ONE workgroup:
    // every dependent trip to memory and every serialised LDS atomic shows in full (one dependent load per query and
    // pass, one returning atomic per query: 24 us, more than half of the forward itself).  So: the reference points
    // are read once, coalesced, all loads of a thread in flight together and before the level geometry is read; the
    // group of every query goes to LDS; then a thread takes K CONSECUTIVE queries - consecutive queries of a grid row
    // fall into the same group in runs (16 of a 128-wide grid on a 64-wide map), it sees 2 - 3 runs - and does ONE
    // counting atomic per run (the run's first query adds the run's length; the value returned is its first rank).
    const bool fits = Lq <= kRegQ * kSchedThreads;
    float2 rp[kRegQ];
    if (fits) {
#pragma unroll
        for (int i = 0; i < kRegQ; ++i)        // clamped: the loads are unconditional
            rp[i] = *reinterpret_cast<const float2 *>(ref + (int64_t)min(i * kSchedThreads + (int)threadIdx.x, Lq - 1) * 2);
    }
    const Level lv = read_level(shapes, lsi, 0, S);
    WinHeader *hd = reinterpret_cast<WinHeader *>(ws);
    int *group_off = reinterpret_cast<int *>(ws + sizeof(WinHeader));
    int *cursor = group_off + Gmax + 1;
    int *perm = cursor + Gmax;
    const int ntx = lv.valid ? (lv.W + kTile - 1) / kTile : 1, nty = lv.valid ? (lv.H + kTile - 1) / kTile : 1;
    const int G = ntx * nty;
    const bool ok = lv.valid && G <= Gmax;
    if (threadIdx.x == 0) {
        WinHeader h{};
        h.valid = ok, h.H = lv.H, h.W = lv.W, h.ntx = ntx, h.nty = nty, h.G = ok ? G : 0, h.start = lv.start;
        *hd = h;
    }
    if (!ok) return;
    const float sx = (float)lv.W * (1.f / kTile), sy = (float)lv.H * (1.f / kTile);
    const bool lds = G <= kLdsGroups;
    int *cnt = lds ? s_cnt : cursor;
    for (int i = threadIdx.x; i < G; i += kSchedThreads) cnt[i] = 0;
    __syncthreads();
    const bool regs = lds && fits;
    uint32_t gr[kRegQ];
    const int K = (Lq + kSchedThreads - 1) / kSchedThreads, q0 = (int)threadIdx.x * K;     // phase 2: queries [q0, q0 + K) of this thread
    if (regs) {
        // (K is uniform: the guards below are scalar branches, iterations beyond K cost nothing)
#pragma unroll
        for (int i = 0; i < kRegQ; ++i) {
            const int q = i * kSchedThreads + (int)threadIdx.x;
            if (i < K && q < Lq) s_g[q] = (unsigned short)group_of(rp[i].x, rp[i].y, sx, sy, ntx, nty);
        }
        __syncthreads();
        int gq[kRegQ];
#pragma unroll
        for (int i = 0; i < kRegQ; ++i) gq[i] = (i < K && q0 + i < Lq) ? (int)s_g[q0 + i] : -1;
        int len[kRegQ];                         // queries from i to the end of its run
        len[kRegQ - 1] = 1;
#pragma unroll
        for (int i = kRegQ - 2; i >= 0; --i) len[i] = (i < K && gq[i + 1] == gq[i]) ? len[i + 1] + 1 : 1;
        int base[kRegQ];
#pragma unroll
        for (int i = 0; i < kRegQ; ++i) {       // independent atomics: several in flight
            base[i] = -1;
            if (i < K) {
                const bool head = gq[i] >= 0 && (i == 0 || gq[i - 1] != gq[i]);
                if (head) base[i] = atomicAdd(&s_cnt[gq[i]], len[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < kRegQ; ++i) {
            gr[i] = 0u;
            if (i < K) {
                if (i > 0 && base[i] < 0) base[i] = base[i - 1] + 1;          // behind the head: the next rank
                gr[i] = gq[i] >= 0 ? (uint32_t)gq[i] | ((uint32_t)base[i] << 12) : 0u;     // rank < Lq <= 2^15, group < 2^12
            }
        }
    } else {
        for (int q = threadIdx.x; q < Lq; q += kSchedThreads) {
            const float2 r1 = *reinterpret_cast<const float2 *>(ref + (int64_t)q * 2);
            atomicAdd(&cnt[group_of(r1.x, r1.y, sx, sy, ntx, nty)], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the G counts
    if (G <= 64) {                       // the adapter: one wave, shuffles
        if (threadIdx.x < 64) {
            const int lane = (int)threadIdx.x;
            const int c = lane < G ? cnt[lane] : 0;
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            if (lane < G) {
                group_off[lane] = incl - c;
                cnt[lane] = incl - c;        // the group's first slot
            }
            if (lane == 0) group_off[G] = Lq;          // every query is in exactly one group
        }
    } else {                             // each thread takes a contiguous slice
        const int per = (G + kSchedThreads - 1) / kSchedThreads;
        const int b0 = min((int)threadIdx.x * per, G), b1 = min(b0 + per, G);
        int sum = 0;
        for (int i = b0; i < b1; ++i) sum += cnt[i];
        s_scan[threadIdx.x] = sum;
        __syncthreads();
        const int nact = min(kSchedThreads, G);              // slices beyond the groups are empty: they add nothing
        for (int d = 1; d < nact; d <<= 1) {
            const int v = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
            __syncthreads();
            s_scan[threadIdx.x] += v;
            __syncthreads();
        }
        int run = s_scan[threadIdx.x] - sum;
        for (int i = b0; i < b1; ++i) {
            const int c = cnt[i];
            group_off[i] = run;
            cnt[i] = run;                       // the group's first slot (fast path) / its write cursor
            run += c;
        }
        if (threadIdx.x == 0) group_off[G] = Lq;            // every query is in exactly one group
    }
    __syncthreads();
    if (regs) {
#pragma unroll
        for (int i = 0; i < kRegQ; ++i)
            if (i < K) {
                if (q0 + i < Lq) perm[s_cnt[gr[i] & 0xFFFu] + (int)(gr[i] >> 12)] = q0 + i;
            }
    } else {
        for (int q = threadIdx.x; q < Lq; q += kSchedThreads) {
            const float2 r1 = *reinterpret_cast<const float2 *>(ref + (int64_t)q * 2);
            perm[atomicAdd(&cnt[group_of(r1.x, r1.y, sx, sy, ntx, nty)], 1)] = q;
        }
    }
}

// os / ls: elements between the offsets / logits of consecutive (n, q, m) rows (contiguous tensors: 8 and 4; the module's
// interleaved fp32 [offsets | logits] rows: 12 and 12)
template <typename VT, typename PT>
__global__ __launch_bounds__(kWinThreads) void msda_fused_fwd_win(
    const VT *__restrict__ value, const PT *__restrict__ off, const PT *__restrict__ logit, int64_t os, int64_t ls,
    const float *__restrict__ ref, const unsigned char *__restrict__ ws, int Gmax, int halo, int64_t S, int M, int N, int64_t Lq,
    VT *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char Vs[];
    constexpr int ROWB = kD * (int)sizeof(VT);
    constexpr int PCS = ROWB / 16;
    const WinHeader &hd = *reinterpret_cast<const WinHeader *>(ws);
    const int *group_off = reinterpret_cast<const int *>(ws + sizeof(WinHeader));
    const int *perm = group_off + Gmax + 1 + Gmax;
    const int H = hd.H, W = hd.W, G = hd.G;
    const int64_t start = hd.start;
    const int64_t items = (int64_t)N * G * M;
    // work items (n, group, head), heads of a group adjacent; XCD b % 8 walks a contiguous eighth
    const int64_t per = (items + 7) / 8;
    for (int64_t it = (int64_t)(blockIdx.x / 8); it < per; it += gridDim.x / 8) {
        const int64_t blk = (int64_t)(blockIdx.x % 8) * per + it;
        if (blk >= items) break;
        const int m = (int)(blk % M);
        const int g = (int)((blk / M) % G);
        const int64_t n = blk / M / G;
        const int gy = g / hd.ntx, gx = g - gy * hd.ntx;
        // the window: the tile + halo + 1 pixels on every side, clipped to the map
        const int wy0 = max(gy * kTile - halo - 1, 0), wx0 = max(gx * kTile - halo - 1, 0);
        const int wh = min(gy * kTile + kTile + halo + 1, H) - wy0, ww = min(gx * kTile + kTile + halo + 1, W) - wx0;
        const int64_t stride = (int64_t)M * kD;
        const VT *vmap = value + (n * S + start) * stride + m * kD;
        __syncthreads();                    // the previous item's readers are done with Vs
        // ---- the window: whole 64 / 128-byte rows, 16 bytes per lane and load
        for (int i = threadIdx.x; i < wh * ww * PCS; i += kWinThreads) {
            const int wp = i / PCS, pc = i - wp * PCS;
            const int wy = wp / ww, wx = wp - wy * ww;
            *reinterpret_cast<uint4 *>(Vs + wp * ROWB + 16 * pc) = *reinterpret_cast<const uint4 *>(
                reinterpret_cast<const unsigned char *>(vmap + ((int64_t)(wy0 + wy) * W + wx0 + wx) * stride) + 16 * pc);
        }
        __syncthreads();
        const int beg = group_off[g], end = group_off[g + 1];
        for (int idx = beg + threadIdx.x; idx < end; idx += kWinThreads) {
            const int64_t q = perm[idx];
            const int64_t row = (n * Lq + q) * M + m;
            // ---- the row's operands: 4 offsets (x, y), 4 logits, the reference point
            uint32_t ow[kP * 2 * sizeof(PT) / 4], lw[kP * sizeof(PT) / 4];
            {
                const uint4 *op = reinterpret_cast<const uint4 *>(off + row * os);
#pragma unroll
                for (int i = 0; i < (int)(kP * 2 * sizeof(PT) / 16); ++i) {
                    const uint4 v = op[i];
                    ow[4 * i] = v.x, ow[4 * i + 1] = v.y, ow[4 * i + 2] = v.z, ow[4 * i + 3] = v.w;
                }
                const uint2 *lp = reinterpret_cast<const uint2 *>(logit + row * ls);
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
                const float hs = inside ? h_im : 0.f, wsx = inside ? w_im : 0.f;
                const float hf = floorf(hs), wf = floorf(wsx);
                const int y0 = (int)hf, x0 = (int)wf;
                const float lh = hs - hf, lwt = wsx - wf, hh = 1.f - lh, hw = 1.f - lwt;
                const float cw[4] = {hh * hw, hh * lwt, lh * hw, lh * lwt};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
                    const bool valid = inside && yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1;
                    const int wy = yy - wy0, wx = xx - wx0;
                    const bool inwin = valid && (unsigned)wy < (unsigned)wh && (unsigned)wx < (unsigned)ww;
                    const float wgt = a[p] * cw[c];
                    axpy_row<VT>(acc, inwin ? wgt : 0.f, Vs + (inwin ? wy * ww + wx : 0) * ROWB);
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
}

int64_t win_gmax(int64_t S) { return S / 8 + 2; }        // groups of 8 x 8 pixels of an H x W map with H * W <= S: at W = 1, H / 8 + 1

template <typename VT, typename PT>
int launch(const void *value, const void *off, const void *logit, int64_t os, int64_t ls, const float *ref, const int64_t *shapes,
           const int64_t *lsi, int64_t N, int64_t S, int64_t M, int64_t Lq, int halo, void *ws, bool ws_ready, void *out,
           hipStream_t st) {
    const int Gmax = (int)win_gmax(S);
    if (!ws_ready) {
        constexpr int sched_lds = (kLdsGroups + kSchedThreads) * 4 + kRegQ * kSchedThreads * 2;
        if (int rc = allow_dynamic_lds((const void *)msda_win_schedule, sched_lds, "msda window schedule")) return rc;
        hipLaunchKernelGGL(msda_win_schedule, dim3(1), dim3(kSchedThreads), sched_lds, st, ref, shapes, lsi, (int)Lq, S, Gmax,
                           (unsigned char *)ws);
        if (int rc = check_launch("msda window schedule launch")) return rc;
    }
    const int side = kTile + 2 * (halo + 1);
    const int smem = side * side * kD * (int)sizeof(VT);
    if (smem > 64 * 1024) return fail(VAH_E_SHAPE, "msda fused forward (windows): a halo of %d pixels needs too much LDS", halo);
    if (int rc = allow_dynamic_lds((const void *)msda_fused_fwd_win<VT, PT>, smem, "msda fused forward (windows)")) return rc;
    // as many workgroups as there can be items, up to a few per CU slot: the kernel loops over the actual items
    int64_t grid = N * M * Gmax;
    if (grid > 16384) grid = 16384;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL((msda_fused_fwd_win<VT, PT>), dim3((unsigned)grid), dim3(kWinThreads), smem, st, (const VT *)value,
                       (const PT *)off, (const PT *)logit, os, ls, ref, (const unsigned char *)ws, Gmax, halo, S, (int)M, (int)N, Lq,
                       (VT *)out);
    return check_launch("msda fused forward (windows) launch");
}

}  // namespace
}  // namespace vah

extern "C" {

int64_t vah_msda_win_ws_bytes(int64_t S, int64_t Lq) {
    vah::clear_error();
    if (S < 1 || Lq < 0 || Lq > (1 << 18) || S >= (1 << 24)) {
        vah::fail(VAH_E_UNSUPPORTED, "vah_msda_win_ws_bytes: the window forward takes up to 2^18 queries of a level of up to 2^24 pixels");
        return -1;
    }
    const int64_t g = vah::win_gmax(S);
    return ((int64_t)sizeof(vah::WinHeader) + (2 * g + 1 + Lq) * 4 + 255) / 256 * 256;
}

int vah_msda_fused_forward_win(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi, const void *offsets,
                               const void *logits, int param_dtype, int64_t offsets_stride, int64_t logits_stride, const float *ref,
                               int64_t N, int64_t S, int64_t M, int64_t D, int64_t Lq, int64_t P, int64_t halo, void *ws,
                               int64_t ws_bytes, int ws_holds_schedule, void *out, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_fused_forward_win";
    if (N < 0 || S < 1 || M < 1 || Lq < 0 || halo < 0 || halo > 32 || M * D >= (1LL << 31)) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (D != kD || P != kP) return fail(VAH_E_UNSUPPORTED, "%s: needs D == 32, P == 4 (one level)", fn);
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !offsets || !logits || !ref || !ws || !out) return fail(VAH_E_NULL, "%s: null pointer", fn);
    const int64_t need = vah_msda_win_ws_bytes(S, Lq);
    if (need < 0) return VAH_E_UNSUPPORTED;
    if (ws_bytes < need) return fail(VAH_E_SHAPE, "%s: workspace too small (%lld < %lld)", fn, (long long)ws_bytes, (long long)need);
    const int64_t ps = param_dtype == 1 ? 2 : 4;
    const int64_t os = offsets_stride ? offsets_stride : kP * 2, ls = logits_stride ? logits_stride : kP;
    if (((uintptr_t)value | (uintptr_t)out | (uintptr_t)offsets | (uintptr_t)ws) % 16 || ((uintptr_t)logits | (uintptr_t)ref) % 8 ||
        (os * ps) % 16 || (ls * ps) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    hipStream_t st = (hipStream_t)stream;
    const int64_t vs = value_dtype == 1 ? 2 : 4;
    LaunchScope scope("msda_fused_fwd", vs * (N * S * M * D + N * Lq * M * D) + ps * 3 * N * Lq * M * P, st,
                      4 * (N * S * M * D + 3 * N * Lq * M * P + N * Lq * M * D));
#define VAH_CASE(VT, VC, PT, PC)                                                                                     \
    if (value_dtype == VC && param_dtype == PC)                                                                      \
        return launch<VT, PT>(value, offsets, logits, os, ls, ref, shapes, lsi, N, S, M, Lq, (int)halo, ws, ws_holds_schedule != 0, out, st)
    VAH_CASE(float, 0, float, 0);
    VAH_CASE(__bf16, 1, __bf16, 1);
    VAH_CASE(__bf16, 1, float, 0);
    VAH_CASE(float, 0, __bf16, 1);
#undef VAH_CASE
    return fail(VAH_E_UNSUPPORTED, "%s: dtype codes must be 0 (f32) or 1 (bf16)", fn);
}

}  // extern "C"
