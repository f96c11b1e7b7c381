// Multi-scale deformable attention for gfx950 (MI355X): forward gather + backward scatter.
//
// Behavioural spec (NOT a translation): /root/reference/detection/ops/src/cuda/
// ms_deform_im2col_cuda.cuh:33-84 (4-corner bilinear read, per-corner zero padding),
// :237-299 (forward), :87-159 + :301-403 (backward).  See include/vitadapter_hip.h for the ABI.
//
// Design (wave64, 128-byte value rows when D = 32):
//   * one (batch, query, head) "row" of the output owns D channels.  Forward: a row is
//     handled by D/4 adjacent lanes, each lane owning one float4 of the channel vector, so a
//     wave-wide global_load_dwordx4 fetches 64/(D/4) complete, fully coalesced corner rows
//     (8 x 128 B at D = 32).  loc / attn are read once per lane group (same-address lanes
//     broadcast in the TA), not once per channel as in the reference.
//   * all four corner loads of P points are issued back to back (invalid corners are
//     clamped to row 0 of the level and zeroed by select, so loads are unconditional and
//     nothing diverges), giving 16 outstanding 16-byte loads per lane at P = 4.
//   * blockIdx is remapped so that each XCD (blockIdx % 8) walks one contiguous eighth of
//     the rows: neighbouring queries share corner rows, and sharing only pays inside one L2.
//   * backward: one lane per channel (D lanes per row) so that each global_atomic_add_f32
//     wave-instruction covers whole 128-byte rows (the shape the memory-side atomic units run
//     at full rate for); loc / attn gradients are reduced across the D lanes with DPP/shuffle
//     butterflies - no LDS, no barriers, no serial 32-term sum.
//   * spatial_shapes / level_start_index stay on the device (scalar loads), as in the
//     reference: no host sync anywhere.
#include "msda_common.h"
#include "msda_internal.h"

#include <cstdlib>

namespace vah {
namespace {

using namespace vah::msda;



// ---------------------------------------------------------------------------------------
// Forward, f32, D = 4*LANES: LANES lanes x float4 per row.
// ---------------------------------------------------------------------------------------
// QMAJOR: consecutive lane groups take consecutive QUERIES of one head (work index = (n, m, q))
// instead of consecutive heads of one query.  Neighbouring queries sample overlapping corner rows,
// so the 8 rows a wave instruction asks for collapse to fewer distinct cache lines in the TA and
// neighbouring waves re-hit L1: less L2 -> L1 traffic, which is what bounds this kernel.
struct TapF {           // one sampling tap as the channel lanes of a row read it back from LDS
    int row[4];         // token index of the corner inside its level (0 for an invalid corner)
    float w[4];         // attention weight x bilinear weight of the corner (0 for an invalid corner)
};

// Phase 1: every sampling tap of the workgroup's rows is computed ONCE by one thread and handed to
// the LANES channel lanes of its row through LDS (dynamic, ROWS * L * P entries of 32 bytes); before,
// each channel lane redid the floor / index arithmetic of all L*P taps of its row, and that
// redundant VALU work - not the gather - was half of the kernel's time on the adapter shapes.
template <int LANES, int PU, bool QMAJOR>
__global__ __launch_bounds__(kBlock) void msda_fwd_vec4(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ attn, int64_t S, int M, int L, int64_t Lq, int P,
    int64_t total_rows, int64_t nblocks, float *__restrict__ out) {
    constexpr int D = 4 * LANES;
    constexpr int ROWS = kBlock / LANES;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    TapF *s_tap = reinterpret_cast<TapF *>(s_raw);
    const int64_t blk = xcd_chunked_block(nblocks);
    if (blk >= nblocks) return;
    const int LP = L * P;
    auto row_of = [&](int64_t work, int &m, int64_t &n) -> int64_t {
        if (QMAJOR) {
            const int64_t q = work % Lq;
            m = (int)((work / Lq) % M);
            n = work / Lq / M;
            return (n * Lq + q) * M + m;
        }
        m = (int)(work % M);
        n = work / M / Lq;
        return work;
    };
    for (int i = threadIdx.x; i < ROWS * LP; i += kBlock) {
        const int rl = i / LP, sidx = i - rl * LP, l = sidx / P;
        const int64_t work = blk * ROWS + rl;
        TapF tl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tl.row[k] = 0;
            tl.w[k] = 0.f;
        }
        const Level lv = read_level(shapes, lsi, l, S);
        if (work < total_rows && lv.valid) {
            int m;
            int64_t n;
            const int64_t row = row_of(work, m, n);
            const float2 xy = *reinterpret_cast<const float2 *>(loc + (row * LP + sidx) * 2);
            const float a = attn[row * LP + sidx];
            const Tap<float> t = make_tap<float>(xy.x, xy.y, lv.H, lv.W);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tl.row[k] = t.row[k];
                tl.w[k] = t.ok[k] ? t.cw[k] * a : 0.f;
            }
        }
        s_tap[i] = tl;
    }
    __syncthreads();
    const int sub = threadIdx.x % LANES, rl = threadIdx.x / LANES;
    const int64_t work = blk * ROWS + rl;
    if (work >= total_rows) return;
    int m;
    int64_t n;
    const int64_t row = row_of(work, m, n);
    const int64_t stride = (int64_t)M * D;                       // floats per token
    const float *vhead = value + n * S * stride + m * D + sub * 4;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
        const Level lv = read_level(shapes, lsi, l, S);
        if (!lv.valid) continue;
        const float *vl = vhead + lv.start * stride;
        for (int p0 = 0; p0 < P; p0 += PU) {
            TapF t[PU];
            float4 v[PU][4];
#pragma unroll
            for (int u = 0; u < PU; ++u) t[u] = s_tap[rl * LP + l * P + p0 + u];
#pragma unroll
            for (int u = 0; u < PU; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    v[u][k] = *reinterpret_cast<const float4 *>(vl + (int64_t)t[u].row[k] * stride);
#pragma unroll
            for (int u = 0; u < PU; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float c = t[u].w[k];
                    acc.x += c * v[u][k].x;
                    acc.y += c * v[u][k].y;
                    acc.z += c * v[u][k].z;
                    acc.w += c * v[u][k].w;
                }
        }
    }
    *reinterpret_cast<float4 *>(out + row * D + sub * 4) = acc;
}

// ---------------------------------------------------------------------------------------
// Backward, f32, one lane per channel, D in {16, 32, 64}.
// ---------------------------------------------------------------------------------------
// Sum over the D lanes of a row.  kResultLane<D> is the lane (within the row) that holds it.
template <int D>
__device__ __forceinline__ float group_sum(float x) {
    if constexpr (D == 32) return dpp_sum32_hi(x);
    if constexpr (D == 16) return dpp_sum16(x);
#pragma unroll
    for (int o = D / 2; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
template <int D>
constexpr int kResultLane = (D == 32) ? 16 : 0;

template <int D, int PU, bool SCATTER = true>
__global__ __launch_bounds__(kBlock) void msda_bwd_lanec(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ attn, const float *__restrict__ grad_out, int64_t S, int M, int L,
    int64_t Lq, int P, int64_t total_rows, int64_t nblocks, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn) {
    constexpr int ROWS = kBlock / D;
    const int64_t blk = xcd_chunked_block(nblocks);
    if (blk >= nblocks) return;
    const int c = threadIdx.x % D;
    const int64_t row = blk * ROWS + threadIdx.x / D;
    if (row >= total_rows) return;     // whole D-lane groups leave together
    const int m = (int)(row % M);
    const int64_t n = row / M / Lq;
    const int64_t stride = (int64_t)M * D;
    const int64_t head_off = n * S * stride + m * D + c;
    const float *lp = loc + row * (int64_t)(L * P) * 2;
    const float *ap = attn + row * (int64_t)(L * P);
    float *glp = grad_loc + row * (int64_t)(L * P) * 2;
    float *gap = grad_attn + row * (int64_t)(L * P);
    const float g = grad_out[row * D + c];

    for (int l = 0; l < L; ++l) {
        const Level lv = read_level(shapes, lsi, l, S);
        if (!lv.valid) {
            if (c == 0)
                for (int p = 0; p < P; ++p) {
                    glp[2 * (l * P + p)] = 0.f;
                    glp[2 * (l * P + p) + 1] = 0.f;
                    gap[l * P + p] = 0.f;
                }
            continue;
        }
        const float *vl = value + head_off + lv.start * stride;
        float *gvl = grad_value + head_off + lv.start * stride;
        for (int p0 = 0; p0 < P; p0 += PU) {
            Tap<float> t[PU];
            float a[PU];
            float v[PU][4];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int s = l * P + p0 + u;
                const float2 xy = *reinterpret_cast<const float2 *>(lp + 2 * s);
                a[u] = ap[s];
                t[u] = make_tap<float>(xy.x, xy.y, lv.H, lv.W);
            }
#pragma unroll
            for (int u = 0; u < PU; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = vl[(int64_t)t[u].row[k] * stride];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int s = l * P + p0 + u;
                const float tv = g * a[u];
                float vv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vv[k] = t[u].ok[k] ? v[u][k] : 0.f;
                    if (SCATTER && t[u].ok[k]) atomicAdd(gvl + (int64_t)t[u].row[k] * stride, t[u].cw[k] * tv);
                }
                const float gh = t[u].hw * (vv[2] - vv[0]) + t[u].lw * (vv[3] - vv[1]);
                const float gw = t[u].hh * (vv[1] - vv[0]) + t[u].lh * (vv[3] - vv[2]);
                const float val = t[u].cw[0] * vv[0] + t[u].cw[1] * vv[1] + t[u].cw[2] * vv[2] +
                                  t[u].cw[3] * vv[3];
                const float pa = group_sum<D>(g * val);
                const float pw = group_sum<D>((float)lv.W * gw * tv);
                const float ph = group_sum<D>((float)lv.H * gh * tv);
                if (c == kResultLane<D>) {
                    *reinterpret_cast<float2 *>(glp + 2 * s) = make_float2(pw, ph);
                    gap[s] = pa;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// d(loc), d(attn) only (no grad_value scatter: that is the tile pass of msda_tile.hip), f32, D = 32:
// 8 lanes x float4 per row as in the forward - 16-byte corner loads, the taps of the workgroup's rows
// computed once by one thread each and handed over through LDS, the three per-sample sums reduced
// over the 8 lanes of a row with DPP adds.  (msda_bwd_lanec with its scatter compiled out does the
// same with one channel per lane and 4-byte loads: 204 us on the extractor call of BASELINE
// configs[2], against 69 us for the forward's gather of the same rows.)
// Spec: cuh:301-403 minus ms_deform_attn_col2im_bilinear's atomics; grad_loc = (W, H) * (...), cuh:157-158.
// ---------------------------------------------------------------------------------------
struct TapG {           // 32 bytes: token index of each corner inside its level (-1 invalid), fractions, weight
    int row[4];
    float lh, lw, a;
    int level;
};

__device__ __forceinline__ float sum8_dpp(float x) {
    x += dpp_mov<0xB1, 0xF>(x);     // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E, 0xF>(x);     // quad_perm [2,3,0,1]
    x += dpp_mov<0x141, 0xF>(x);    // row_half_mirror: lanes 0-7 <-> 7-0 within each 8
    return x;                       // every lane of the 8-lane group holds the sum
}

template <int PU>
__global__ __launch_bounds__(kBlock) void msda_bwd_taps_vec4(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attn, const float *__restrict__ grad_out,
    int64_t S, int M, int L, int64_t Lq, int P, int64_t total_rows, int64_t nblocks,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn) {
    constexpr int ROWS = kBlock / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    TapG *s_tap = reinterpret_cast<TapG *>(s_raw);
    const int64_t blk = xcd_chunked_block(nblocks);
    if (blk >= nblocks) return;
    const int LP = L * P;
    auto row_of = [&](int64_t work, int &m, int64_t &n) -> int64_t {      // query-major work order (see the forward)
        const int64_t q = work % Lq;
        m = (int)((work / Lq) % M);
        n = work / Lq / M;
        return (n * Lq + q) * M + m;
    };
    for (int i = threadIdx.x; i < ROWS * LP; i += kBlock) {
        const int rl = i / LP, sidx = i - rl * LP, l = sidx / P;
        const int64_t work = blk * ROWS + rl;
        TapG tl;
        tl.row[0] = tl.row[1] = tl.row[2] = tl.row[3] = -1;
        tl.lh = tl.lw = tl.a = 0.f;
        tl.level = l;
        const Level lv = read_level(shapes, lsi, l, S);
        if (work < total_rows && lv.valid) {
            int m;
            int64_t n;
            const int64_t row = row_of(work, m, n);
            const float2 xy = *reinterpret_cast<const float2 *>(loc + (row * LP + sidx) * 2);
            const Tap<float> t = make_tap<float>(xy.x, xy.y, lv.H, lv.W);
#pragma unroll
            for (int k = 0; k < 4; ++k) tl.row[k] = t.ok[k] ? t.row[k] : -1;
            tl.lh = t.lh;
            tl.lw = t.lw;
            tl.a = attn[row * LP + sidx];
        }
        s_tap[i] = tl;
    }
    __syncthreads();
    const int sub = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int64_t work = blk * ROWS + rl;
    if (work >= total_rows) return;          // whole 8-lane groups leave together (after the barrier)
    int m;
    int64_t n;
    const int64_t row = row_of(work, m, n);
    const int64_t stride = (int64_t)M * 32;
    const float *vhead = value + n * S * stride + m * 32 + sub * 4;
    const float4 g = *reinterpret_cast<const float4 *>(grad_out + row * 32 + sub * 4);
    float *glp = grad_loc + row * LP * 2;
    float *gap = grad_attn + row * LP;
    for (int s0 = 0; s0 < LP; s0 += PU) {
        const TapG *tp = s_tap + rl * LP + s0;
        float4 v[PU][4];
        // P is a multiple of PU: the PU samples of this step belong to ONE level (scalar geometry loads)
        const Level lvl = read_level(shapes, lsi, s0 / P, S);
        Level lv[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            lv[u] = lvl;
            const float *vl = vhead + (lv[u].valid ? lv[u].start : 0) * stride;
            const int4 rw = *reinterpret_cast<const int4 *>(tp[u].row);
            const int r4[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // unconditional (clamped) load + select: a load under `if (valid)` is waited on alone
                const float4 x = *reinterpret_cast<const float4 *>(vl + (int64_t)max(r4[k], 0) * stride);
                const bool ok = r4[k] >= 0;
                v[u][k] = make_float4(ok ? x.x : 0.f, ok ? x.y : 0.f, ok ? x.z : 0.f, ok ? x.w : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const float lh = tp[u].lh, lw = tp[u].lw, hh = 1.f - lh, hw = 1.f - lw, a = tp[u].a;
            auto dot4 = [&](const float4 &x) { return g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w; };
            const float d0 = dot4(v[u][0]), d1 = dot4(v[u][1]), d2 = dot4(v[u][2]), d3 = dot4(v[u][3]);
            const float val = hh * hw * d0 + hh * lw * d1 + lh * hw * d2 + lh * lw * d3;
            const float gh = hw * (d2 - d0) + lw * (d3 - d1);
            const float gw = hh * (d1 - d0) + lh * (d3 - d2);
            const float pa = sum8_dpp(val);
            const float pw = sum8_dpp((float)lv[u].W * gw * a);
            const float ph = sum8_dpp((float)lv[u].H * gh * a);
            if (sub == ((s0 + u) & 7)) {
                *reinterpret_cast<float2 *>(glp + 2 * (s0 + u)) = lv[u].valid ? make_float2(pw, ph) : make_float2(0.f, 0.f);
                gap[s0 + u] = lv[u].valid ? pa : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Generic fallback (any D, f32 / f64): one wave per row, lanes stride over channels.
// Used for every D the fast paths do not cover and for the fp64 gradcheck path
// (the reference dispatches float and double only, ms_deform_attn_cuda.cu:64,134).
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void msda_fwd_generic(
    const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ attn,
    int64_t S, int M, int D, int L, int64_t Lq, int P, int64_t total_rows, T *__restrict__ out) {
    const int lane = threadIdx.x % kWave;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (row >= total_rows) return;
    const int m = (int)(row % M);
    const int64_t n = row / M / Lq;
    const int64_t stride = (int64_t)M * D;
    const T *vhead = value + n * S * stride + (int64_t)m * D;
    const T *lp = loc + row * (int64_t)(L * P) * 2;
    const T *ap = attn + row * (int64_t)(L * P);
    for (int c0 = 0; c0 < D; c0 += kWave) {
        const int c = c0 + lane;
        const bool live = c < D;
        const int cc = live ? c : 0;
        T acc = 0;
        for (int l = 0; l < L; ++l) {
            const Level lv = read_level(shapes, lsi, l, S);
            if (!lv.valid) continue;
            const T *vl = vhead + lv.start * stride + cc;
            for (int p = 0; p < P; ++p) {
                const int s = l * P + p;
                const Tap<T> t = make_tap<T>(lp[2 * s], lp[2 * s + 1], lv.H, lv.W);
                T val = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const T x = vl[(int64_t)t.row[k] * stride];
                    val += t.cw[k] * (t.ok[k] ? x : (T)0);
                }
                acc += val * ap[s];
            }
        }
        if (live) out[row * D + c] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void msda_bwd_generic(
    const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ attn,
    const T *__restrict__ grad_out, int64_t S, int M, int D, int L, int64_t Lq, int P,
    int64_t total_rows, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_attn) {
    const int lane = threadIdx.x % kWave;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (row >= total_rows) return;     // whole waves leave together
    const int m = (int)(row % M);
    const int64_t n = row / M / Lq;
    const int64_t stride = (int64_t)M * D;
    const int64_t head_off = n * S * stride + (int64_t)m * D;
    const T *lp = loc + row * (int64_t)(L * P) * 2;
    const T *ap = attn + row * (int64_t)(L * P);
    T *glp = grad_loc + row * (int64_t)(L * P) * 2;
    T *gap = grad_attn + row * (int64_t)(L * P);
    const T *g = grad_out + row * D;
    for (int l = 0; l < L; ++l) {
        const Level lv = read_level(shapes, lsi, l, S);
        for (int p = 0; p < P; ++p) {
            const int s = l * P + p;
            T pa = 0, pw = 0, ph = 0;
            if (lv.valid) {
                const Tap<T> t = make_tap<T>(lp[2 * s], lp[2 * s + 1], lv.H, lv.W);
                const T a = ap[s];
                const T *vl = value + head_off + lv.start * stride;
                T *gvl = grad_value + head_off + lv.start * stride;
                for (int c = lane; c < D; c += kWave) {
                    const T top = g[c];
                    const T tv = top * a;
                    T vv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int64_t off = (int64_t)t.row[k] * stride + c;
                        vv[k] = t.ok[k] ? vl[off] : (T)0;
                        if (t.ok[k]) atomicAdd(gvl + off, t.cw[k] * tv);
                    }
                    const T gh = t.hw * (vv[2] - vv[0]) + t.lw * (vv[3] - vv[1]);
                    const T gw = t.hh * (vv[1] - vv[0]) + t.lh * (vv[3] - vv[2]);
                    const T val = t.cw[0] * vv[0] + t.cw[1] * vv[1] + t.cw[2] * vv[2] +
                                  t.cw[3] * vv[3];
                    pa += top * val;
                    pw += (T)lv.W * gw * tv;
                    ph += (T)lv.H * gh * tv;
                }
            }
            pa = wave_sum<T>(pa);
            pw = wave_sum<T>(pw);
            ph = wave_sum<T>(ph);
            if (lane == 0) {
                glp[2 * s] = pw;
                glp[2 * s + 1] = ph;
                gap[s] = pa;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct Dims {
    int64_t N, S, M, D, L, Lq, P;
    int64_t rows;        // N*Lq*M
};

int check_dims(const char *fn, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq,
               int64_t P, Dims *d) {
    if (N < 0 || S < 0 || M < 1 || D < 1 || L < 1 || Lq < 0 || P < 1)
        return fail(VAH_E_SHAPE, "%s: bad dims N=%lld S=%lld M=%lld D=%lld L=%lld Lq=%lld P=%lld",
                    fn, (long long)N, (long long)S, (long long)M, (long long)D, (long long)L,
                    (long long)Lq, (long long)P);
    const int64_t lim = (int64_t)1 << 31;
    if (M >= lim || D >= lim || L * P >= lim || M * D >= lim)
        return fail(VAH_E_SHAPE, "%s: M, D or L*P does not fit 32-bit index math", fn);
    // per-level row index and (row * M*D) stay in int64 inside the kernels; total element
    // counts must fit int64 - check the largest product with headroom.
    const long double big = (long double)N * (long double)(S > Lq * L * P * 2 ? S : Lq * L * P * 2) *
                            (long double)M * (long double)D;
    if (big > 9.0e18L) return fail(VAH_E_SHAPE, "%s: tensor too large", fn);
    *d = Dims{N, S, M, D, L, Lq, P, N * Lq * M};
    return VAH_OK;
}

inline int64_t fwd_bytes(const Dims &d, int esz) {
    return esz * (d.N * d.S * d.M * d.D + 3 * d.N * d.Lq * d.M * d.L * d.P + d.N * d.Lq * d.M * d.D);
}
inline int64_t bwd_bytes(const Dims &d, int esz) {
    return esz * (2 * d.N * d.S * d.M * d.D + 6 * d.N * d.Lq * d.M * d.L * d.P + d.N * d.Lq * d.M * d.D);
}

inline bool aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

template <int LANES>
int launch_fwd_vec4(const Dims &d, const float *value, const int64_t *shapes, const int64_t *lsi,
                    const float *loc, const float *attn, float *out, hipStream_t st) {
    constexpr int ROWS = kBlock / LANES;
    const int64_t nblocks = (d.rows + ROWS - 1) / ROWS;
    const int64_t grid = (nblocks + 7) / 8 * 8;
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda forward: grid too large");
    const size_t smem = (size_t)ROWS * d.L * d.P * sizeof(TapF);         // the workgroup's tap table
    static const bool qmajor = [] {
        const char *e = getenv("VAH_MSDA_FWD_QMAJOR");
        return e ? atoi(e) != 0 : true;
    }();
#define VAH_FWD(PU)                                                                               \
    do {                                                                                          \
        if (qmajor)                                                                               \
            hipLaunchKernelGGL((msda_fwd_vec4<LANES, PU, true>), dim3((unsigned)grid), dim3(kBlock), smem, \
                               st, value, shapes, lsi, loc, attn, d.S, (int)d.M, (int)d.L, d.Lq,   \
                               (int)d.P, d.rows, nblocks, out);                                    \
        else                                                                                      \
            hipLaunchKernelGGL((msda_fwd_vec4<LANES, PU, false>), dim3((unsigned)grid), dim3(kBlock), smem, \
                               st, value, shapes, lsi, loc, attn, d.S, (int)d.M, (int)d.L, d.Lq,   \
                               (int)d.P, d.rows, nblocks, out);                                    \
    } while (0)
    if (d.P % 4 == 0) VAH_FWD(4);
    else if (d.P % 2 == 0) VAH_FWD(2);
    else VAH_FWD(1);
#undef VAH_FWD
    return check_launch("msda forward launch");
}

template <int D, bool SCATTER = true>
int launch_bwd_lanec(const Dims &d, const float *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *attn, const float *gout, float *gv, float *gl,
                     float *ga, hipStream_t st) {
    constexpr int ROWS = kBlock / D;
    const int64_t nblocks = (d.rows + ROWS - 1) / ROWS;
    const int64_t grid = (nblocks + 7) / 8 * 8;
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda backward: grid too large");
#define VAH_BWD(PU)                                                                             \
    hipLaunchKernelGGL((msda_bwd_lanec<D, PU, SCATTER>), dim3((unsigned)grid), dim3(kBlock), 0, st, \
                       value, shapes, lsi, loc, attn, gout, d.S, (int)d.M, (int)d.L, d.Lq,      \
                       (int)d.P, d.rows, nblocks, gv, gl, ga)
    if (d.P % 4 == 0) VAH_BWD(4);
    else if (d.P % 2 == 0) VAH_BWD(2);
    else VAH_BWD(1);
#undef VAH_BWD
    return check_launch("msda backward launch");
}

template <typename T>
int forward_impl(const char *fn, const T *value, const int64_t *shapes, const int64_t *lsi,
                 const T *loc, const T *attn, int64_t N, int64_t S, int64_t M, int64_t D,
                 int64_t L, int64_t Lq, int64_t P, T *out, void *stream) {
    clear_error();
    Dims d;
    if (int rc = check_dims(fn, N, S, M, D, L, Lq, P, &d)) return rc;
    if (d.rows == 0) return VAH_OK;                       // empty batch / no queries
    if (!value || !shapes || !lsi || !loc || !attn || !out)
        return fail(VAH_E_NULL, "%s: null pointer argument", fn);
    if (S < 1) return fail(VAH_E_SHAPE, "%s: S must be >= 1 when there are queries", fn);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope(sizeof(T) == 4 ? "msda_fwd_f32" : "msda_fwd_f64", fwd_bytes(d, sizeof(T)), st);
    if constexpr (sizeof(T) == 4) {
        // the vector kernels keep one 32-byte tap per (row, level, point) of the workgroup in LDS
        const bool vec_ok = aligned(value, 16) && aligned(out, 16) && aligned(loc, 8) &&
                            (kBlock * 4 / D) * L * P * 32 <= 48 * 1024;
        if (vec_ok && D == 32) return launch_fwd_vec4<8>(d, value, shapes, lsi, loc, attn, out, st);
        if (vec_ok && D == 16) return launch_fwd_vec4<4>(d, value, shapes, lsi, loc, attn, out, st);
        if (vec_ok && D == 64) return launch_fwd_vec4<16>(d, value, shapes, lsi, loc, attn, out, st);
    }
    const int64_t grid = (d.rows + (kBlock / kWave) - 1) / (kBlock / kWave);
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "%s: grid too large", fn);
    hipLaunchKernelGGL((msda_fwd_generic<T>), dim3((unsigned)grid), dim3(kBlock), 0, st, value,
                       shapes, lsi, loc, attn, S, (int)M, (int)D, (int)L, Lq, (int)P, d.rows, out);
    return check_launch(fn);
}

template <typename T>
int backward_impl(const char *fn, const T *value, const int64_t *shapes, const int64_t *lsi,
                  const T *loc, const T *attn, const T *gout, int64_t N, int64_t S, int64_t M,
                  int64_t D, int64_t L, int64_t Lq, int64_t P, T *gv, T *gl, T *ga, void *stream) {
    clear_error();
    Dims d;
    if (int rc = check_dims(fn, N, S, M, D, L, Lq, P, &d)) return rc;
    if (d.rows == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !loc || !attn || !gout || !gv || !gl || !ga)
        return fail(VAH_E_NULL, "%s: null pointer argument", fn);
    if (S < 1) return fail(VAH_E_SHAPE, "%s: S must be >= 1 when there are queries", fn);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope(sizeof(T) == 4 ? "msda_bwd_f32" : "msda_bwd_f64", bwd_bytes(d, sizeof(T)), st);
    if constexpr (sizeof(T) == 4) {
        const bool vec_ok = aligned(loc, 8) && aligned(gl, 8);
        if (vec_ok && D == 32)
            return launch_bwd_lanec<32>(d, value, shapes, lsi, loc, attn, gout, gv, gl, ga, st);
        if (vec_ok && D == 16)
            return launch_bwd_lanec<16>(d, value, shapes, lsi, loc, attn, gout, gv, gl, ga, st);
        if (vec_ok && D == 64)
            return launch_bwd_lanec<64>(d, value, shapes, lsi, loc, attn, gout, gv, gl, ga, st);
    }
    const int64_t grid = (d.rows + (kBlock / kWave) - 1) / (kBlock / kWave);
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "%s: grid too large", fn);
    hipLaunchKernelGGL((msda_bwd_generic<T>), dim3((unsigned)grid), dim3(kBlock), 0, st, value,
                       shapes, lsi, loc, attn, gout, S, (int)M, (int)D, (int)L, Lq, (int)P, d.rows,
                       gv, gl, ga);
    return check_launch(fn);
}

}  // namespace

int msda_grad_taps_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attn,
                       const float *grad_out, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                       float *grad_loc, float *grad_attn, hipStream_t st) {
    Dims d;
    if (int rc = check_dims("msda_grad_taps_f32", N, S, M, D, L, Lq, P, &d)) return rc;
    if (D != 32) return fail(VAH_E_UNSUPPORTED, "msda_grad_taps_f32: needs D == 32");
    const int64_t smem = (int64_t)(kBlock / 8) * L * P * (int64_t)sizeof(TapG);
    if (P % 4 == 0 && smem <= 48 * 1024 && aligned(value, 16) && aligned(grad_out, 16) && aligned(loc, 8) &&
        aligned(grad_loc, 8)) {
        constexpr int ROWS = kBlock / 8;
        const int64_t nblocks = (d.rows + ROWS - 1) / ROWS;
        const int64_t grid = (nblocks + 7) / 8 * 8;
        if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda_grad_taps_f32: grid too large");
        hipLaunchKernelGGL((msda_bwd_taps_vec4<4>), dim3((unsigned)grid), dim3(kBlock), (size_t)smem, st, value, shapes, lsi, loc,
                           attn, grad_out, d.S, (int)d.M, (int)d.L, d.Lq, (int)d.P, d.rows, nblocks, grad_loc, grad_attn);
        return check_launch("msda_grad_taps_f32");
    }
    // grad_value is never touched with SCATTER off; the value pointer stands in for it
    return launch_bwd_lanec<32, false>(d, value, shapes, lsi, loc, attn, grad_out, const_cast<float *>(value), grad_loc,
                                       grad_attn, st);
}

}  // namespace vah

extern "C" {

int vah_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *attn, int64_t N, int64_t S, int64_t M,
                         int64_t D, int64_t L, int64_t Lq, int64_t P, float *out, void *stream) {
    return vah::forward_impl<float>("vah_msda_forward_f32", value, shapes, lsi, loc, attn, N, S, M,
                                    D, L, Lq, P, out, stream);
}

int vah_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                         const double *loc, const double *attn, int64_t N, int64_t S, int64_t M,
                         int64_t D, int64_t L, int64_t Lq, int64_t P, double *out, void *stream) {
    return vah::forward_impl<double>("vah_msda_forward_f64", value, shapes, lsi, loc, attn, N, S,
                                     M, D, L, Lq, P, out, stream);
}

int vah_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *attn, const float *grad_out, int64_t N,
                          int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                          float *grad_value, float *grad_loc, float *grad_attn, void *stream) {
    return vah::backward_impl<float>("vah_msda_backward_f32", value, shapes, lsi, loc, attn,
                                     grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                                     grad_attn, stream);
}

int vah_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                          const double *loc, const double *attn, const double *grad_out, int64_t N,
                          int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                          double *grad_value, double *grad_loc, double *grad_attn, void *stream) {
    return vah::backward_impl<double>("vah_msda_backward_f64", value, shapes, lsi, loc, attn,
                                      grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                                      grad_attn, stream);
}

}  // extern "C"
