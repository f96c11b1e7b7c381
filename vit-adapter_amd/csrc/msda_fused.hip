// Fused MSDeformAttn core for gfx950: softmax over the L*P attention logits, sampling-location
// arithmetic and the multi-scale bilinear gather in ONE kernel (and their gradients in one more).
//
// Replaces, inside the reference's MSDeformAttn.forward
// (/root/reference/detection/ops/modules/ms_deform_attn.py:108-128):
//     attention_weights = softmax(attention_weights(query).view(N, Lq, M, L*P), -1)
//     sampling_locations = reference_points[:, :, None, :, None, :]
//                          + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
//     output = MSDeformAttnFunction.apply(value, shapes, lsi, sampling_locations, attention_weights, step)
// i.e. ~8 elementwise / softmax kernels, two (N,Lq,M,L,P[,2]) fp32 tensors written and re-read, and
// fp32 casts of the bf16 value / offsets / logits under autocast.  Here the raw Linear outputs
// (fp32 or bf16) and the value tensor (fp32 or bf16: 64-byte rows halve the gathered bytes) are
// read directly; out is written in the value's dtype.  Gather / scatter arithmetic is that of
// msda.hip (spec: ms_deform_im2col_cuda.cuh:33-159,237-403).
//
// Restrictions of this path: D == 32, (L, P) in {(1,4), (3,4), (4,4)}, reference points shared by
// the batch with last dim 2 (the adapter's).  Everything else uses the unfused Function.
#include <cstdlib>
#include <type_traits>

#include "msda_common.h"
#include "msda_internal.h"

namespace vah {
namespace {

using namespace vah::msda;

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;

__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load4(const __bf16 *p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void store4(__bf16 *p, float4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x;
    o[1] = (__bf16)v.y;
    o[2] = (__bf16)v.z;
    o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4 *>(p) = o;
}
__device__ __forceinline__ float2 load2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
__device__ __forceinline__ float2 load2(const __bf16 *p) {
    const bf16x2 v = *reinterpret_cast<const bf16x2 *>(p);
    return make_float2((float)v[0], (float)v[1]);
}
__device__ __forceinline__ void store2(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
__device__ __forceinline__ void store2(__bf16 *p, float a, float b) {
    bf16x2 o;
    o[0] = (__bf16)a;
    o[1] = (__bf16)b;
    *reinterpret_cast<bf16x2 *>(p) = o;
}

// softmax over the LP logits of one (n, q, m) row, in registers
template <typename PT, int LP>
__device__ __forceinline__ void row_softmax(const PT *__restrict__ lg, float (&p)[LP]) {
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < LP; ++s) {
        p[s] = (float)lg[s];
        mx = fmaxf(mx, p[s]);
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < LP; ++s) {
        p[s] = __expf(p[s] - mx);
        sum += p[s];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int s = 0; s < LP; ++s) p[s] *= inv;
}

constexpr int kD = 32;
constexpr float kNoScatter = -2.f;      // near_radius sentinel: the gather kernels scatter nothing

// ---------------------------------------------------------------------------------------
// forward: 8 lanes x 4 channels per row, query-major work order (see msda.hip)
// ---------------------------------------------------------------------------------------
struct TapF {           // one sampling tap as the 8 channel lanes of a row read it back from LDS
    int row[4];         // token index of the corner inside its level (0 for an invalid corner)
    float w[4];         // bilinear weight of the corner (0 for an invalid corner)
};

template <typename VT, typename PT, int L, int P>
__global__ __launch_bounds__(kBlock) void msda_fused_fwd(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const PT *__restrict__ off, const PT *__restrict__ logit, const float *__restrict__ ref,
    int ref_levels, int64_t S, int M, int64_t Lq, int64_t total_rows, int64_t nblocks,
    VT *__restrict__ out) {
    constexpr int LP = L * P;
    constexpr int ROWS = kBlock / 8;
    __shared__ __attribute__((aligned(16))) TapF s_tap[ROWS * LP];
    const int64_t blk = xcd_chunked_block(nblocks);
    if (blk >= nblocks) return;
    // phase 1: every tap of the block's rows once (not once per channel lane), see msda_fused_bwd_vec4
    for (int i = threadIdx.x; i < ROWS * LP; i += kBlock) {
        const int rl = i / LP, sidx = i - rl * LP, l = sidx / P;
        const int64_t w = blk * ROWS + rl;
        TapF tl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tl.row[k] = 0;
            tl.w[k] = 0.f;
        }
        const Level lv = read_level(shapes, lsi, l, S);
        if (w < total_rows && lv.valid) {
            const int64_t q = w % Lq;
            const int64_t rw = (w / Lq / M * Lq + q) * M + (w / Lq) % M;
            const float2 o = load2(off + (rw * LP + sidx) * 2);
            const float2 rp = *reinterpret_cast<const float2 *>(ref + (q * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
            const Tap<float> t = make_tap<float>(rp.x + o.x / (float)lv.W, rp.y + o.y / (float)lv.H, lv.H, lv.W);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tl.row[k] = t.row[k];
                tl.w[k] = t.ok[k] ? t.cw[k] : 0.f;
            }
        }
        s_tap[i] = tl;
    }
    __syncthreads();
    const int sub = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int64_t work = blk * ROWS + rl;
    if (work >= total_rows) return;
    const int64_t q = work % Lq;
    const int m = (int)((work / Lq) % M);
    const int64_t n = work / Lq / M;
    const int64_t row = (n * Lq + q) * M + m;
    const int64_t stride = (int64_t)M * kD;
    const VT *vhead = value + n * S * stride + m * kD + sub * 4;

    float p[LP];
    row_softmax<PT, LP>(logit + row * LP, p);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const Level lv = read_level(shapes, lsi, l, S);
        if (!lv.valid) continue;
        const VT *vl = vhead + lv.start * stride;
        TapF t[P];
        float4 v[P][4];
#pragma unroll
        for (int u = 0; u < P; ++u) t[u] = s_tap[rl * LP + l * P + u];
#pragma unroll
        for (int u = 0; u < P; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = load4(vl + (int64_t)t[u].row[k] * stride);
#pragma unroll
        for (int u = 0; u < P; ++u) {
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float c = t[u].w[k];
                s4.x += c * v[u][k].x;
                s4.y += c * v[u][k].y;
                s4.z += c * v[u][k].z;
                s4.w += c * v[u][k].w;
            }
            const float a = p[l * P + u];
            acc.x += s4.x * a;
            acc.y += s4.y * a;
            acc.z += s4.z * a;
            acc.w += s4.w * a;
        }
    }
    store4(out + row * kD + sub * 4, acc);
}

// ---------------------------------------------------------------------------------------
// backward: one lane per channel (32 lanes per row); grad_value accumulated in fp32 with
// global_atomic_add_f32 on whole 128-byte rows; d(offsets), d(logits) written in PT
// ---------------------------------------------------------------------------------------
// near_radius >= 0 selects the SPLIT mode: samples whose offset is within near_radius pixels (of
// their level) in both axes leave grad_value to msda_fused_bwd_gv below; only the rare "far" samples
// are scattered with atomics here.  near_radius < 0: every sample is scattered here.
template <typename VT, typename PT, int L, int P>
__global__ __launch_bounds__(kBlock) void msda_fused_bwd(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const PT *__restrict__ off, const PT *__restrict__ logit, const float *__restrict__ ref,
    int ref_levels, const VT *__restrict__ grad_out, int64_t S, int M, int64_t Lq, int64_t total_rows,
    int64_t nblocks, float near_radius, float *__restrict__ grad_value, PT *__restrict__ d_off,
    PT *__restrict__ d_logit) {
    // near_radius == kNoScatter: grad_value is left to the tile pass (msda_tile.hip) altogether
    constexpr int LP = L * P;
    constexpr int ROWS = kBlock / kD;
    const int64_t blk = xcd_chunked_block(nblocks);
    if (blk >= nblocks) return;
    const int c = threadIdx.x & 31;
    const int64_t row = blk * ROWS + (threadIdx.x >> 5);
    if (row >= total_rows) return;          // whole 32-lane groups leave together
    const int m = (int)(row % M);
    const int64_t q = (row / M) % Lq;
    const int64_t n = row / M / Lq;
    const int64_t stride = (int64_t)M * kD;
    const int64_t head_off = n * S * stride + m * kD + c;

    float p[LP], ga[LP], gx[LP], gy[LP];
    row_softmax<PT, LP>(logit + row * LP, p);
    const PT *op = off + row * LP * 2;
    const float g = (float)grad_out[row * kD + c];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const Level lv = read_level(shapes, lsi, l, S);
        const float2 rp = *reinterpret_cast<const float2 *>(ref + (q * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
        const int64_t lstart = lv.valid ? lv.start : 0;      // the gathers below are unconditional: stay in bounds
        const VT *vl = value + head_off + lstart * stride;
        float *gvl = grad_value + head_off + lstart * stride;
        Tap<float> t[P];
        float v[P][4];
        bool scatter[P];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const float2 o = load2(op + 2 * (l * P + u));
            scatter[u] = near_radius != kNoScatter && !(fabsf(o.x) <= near_radius && fabsf(o.y) <= near_radius);
            // an invalid level gates every sample off (W = H = 0 would not): use a location that fails
            t[u] = make_tap<float>(lv.valid ? rp.x + o.x / (float)lv.W : -8.f,
                                   lv.valid ? rp.y + o.y / (float)lv.H : -8.f, max(lv.H, 1), max(lv.W, 1));
        }
#pragma unroll
        for (int u = 0; u < P; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {      // clamped load + select (a load under a condition is waited on alone)
                const bool ok = lv.valid && t[u].ok[k];
                const float x = (float)vl[(int64_t)(ok ? t[u].row[k] : 0) * stride];
                v[u][k] = ok ? x : 0.f;
            }
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int s = l * P + u;
            const float tv = g * p[s];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (scatter[u] && lv.valid && t[u].ok[k])
                    atomicAdd(gvl + (int64_t)t[u].row[k] * stride, t[u].cw[k] * tv);
            const float gh = t[u].hw * (v[u][2] - v[u][0]) + t[u].lw * (v[u][3] - v[u][1]);
            const float gw = t[u].hh * (v[u][1] - v[u][0]) + t[u].lh * (v[u][3] - v[u][2]);
            const float val = t[u].cw[0] * v[u][0] + t[u].cw[1] * v[u][1] + t[u].cw[2] * v[u][2] +
                              t[u].cw[3] * v[u][3];
            ga[s] = dpp_sum32_hi(g * val);          // valid on lanes 16..31 of the row
            // d loc = W * gw * tv  and  d off = d loc / W: the level size cancels
            gx[s] = dpp_sum32_hi(gw * tv);
            gy[s] = dpp_sum32_hi(gh * tv);
        }
    }
    // softmax backward + stores: lane 16 + s of the row handles sample s (LP <= 16)
    float dot = 0.f;
#pragma unroll
    for (int s = 0; s < LP; ++s) dot += p[s] * ga[s];
#pragma unroll
    for (int s = 0; s < LP; ++s)
        if (c == 16 + s) {
            d_logit[row * LP + s] = (PT)(p[s] * (ga[s] - dot));
            store2(d_off + (row * LP + s) * 2, gx[s], gy[s]);
        }
}

// Split-mode variant of the backward above: 8 lanes x 4 channels per row (16-byte corner loads as
// in the forward: 4x fewer gather instructions than one channel per lane).  It only makes sense when
// almost no sample scatters here (the far ones do, with 4 strided atomics per lane), i.e. together
// with msda_fused_bwd_gv.
__device__ __forceinline__ float sum8(float x) {
    x += dpp_mov<0xB1, 0xF>(x);     // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E, 0xF>(x);     // quad_perm [2,3,0,1]
    x += dpp_mov<0x141, 0xF>(x);    // row_half_mirror: lanes 0-7 <-> 7-0 within each 8
    return x;                       // every lane of the 8-lane group holds the sum
}

// One sampling tap as the 8 channel lanes of a row read it back from LDS.
struct TapL {
    int row[4];        // token index of the corner inside its level, -1 = invalid corner
    float lh, lw;
    int far;           // 1: outside near_radius (its grad_value goes through atomics here)
    int pad;
};

#ifndef VAH_BWDA_WAVES1
#define VAH_BWDA_WAVES1 6          // waves per SIMD asked for msda_fused_bwd_vec4 at L == 1 / L > 1 (80 / 168 registers):
#endif                             // msda_fused_bwd 170.5 -> 167.7 us per call, A/B of two builds on one box
#ifndef VAH_BWDA_WAVESN
#define VAH_BWDA_WAVESN 3
#endif
template <typename VT, typename PT, int L, int P>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(sizeof(VT) == 2 ? (L == 1 ? VAH_BWDA_WAVES1 : VAH_BWDA_WAVESN) : 1)))   // fp32 values would spill
void msda_fused_bwd_vec4(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const PT *__restrict__ off, const PT *__restrict__ logit, const float *__restrict__ ref,
    int ref_levels, const VT *__restrict__ grad_out, int64_t S, int M, int64_t Lq, int64_t total_rows,
    int64_t nblocks, float near_radius, float *__restrict__ grad_value, PT *__restrict__ d_off,
    PT *__restrict__ d_logit) {
    constexpr int LP = L * P;
    constexpr int ROWS = kBlock / 8;
    __shared__ __attribute__((aligned(16))) TapL s_tap[ROWS * LP];
    const int64_t blk = xcd_chunked_block(nblocks);
    if (blk >= nblocks) return;
    // ---- phase 1: every sampling tap of the block's rows is computed ONCE (the 8 channel lanes of a
    // row used to redo the divisions / floors / index arithmetic of all L*P samples each: the
    // kernel ran 80 us of its 143 us with the value gather switched off)
    for (int i = threadIdx.x; i < ROWS * LP; i += kBlock) {
        const int rl = i / LP, sidx = i - rl * LP, l = sidx / P;
        const int64_t w = blk * ROWS + rl;
        TapL tl;
        tl.row[0] = tl.row[1] = tl.row[2] = tl.row[3] = -1;
        tl.lh = tl.lw = 0.f;
        tl.far = 0;
        tl.pad = 0;
        if (w < total_rows) {
            const int64_t q = w % Lq;
            const int64_t rw = (w / Lq / M * Lq + q) * M + (w / Lq) % M;
            const Level lv = read_level(shapes, lsi, l, S);
            const float2 o = load2(off + (rw * LP + sidx) * 2);
            const float2 rp = *reinterpret_cast<const float2 *>(ref + (q * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
            const Tap<float> t = make_tap<float>(lv.valid ? rp.x + o.x / (float)lv.W : -8.f,
                                                 lv.valid ? rp.y + o.y / (float)lv.H : -8.f, max(lv.H, 1), max(lv.W, 1));
#pragma unroll
            for (int k = 0; k < 4; ++k) tl.row[k] = (lv.valid && t.ok[k]) ? t.row[k] : -1;
            tl.lh = t.lh;
            tl.lw = t.lw;
            tl.far = (lv.valid && near_radius != kNoScatter && !(fabsf(o.x) <= near_radius && fabsf(o.y) <= near_radius)) ? 1 : 0;
        }
        s_tap[i] = tl;
    }
    __syncthreads();
    const int sub = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int64_t work = blk * ROWS + rl;
    if (work >= total_rows) return;          // whole 8-lane groups leave together (after the barrier)
    const int64_t q = work % Lq;
    const int m = (int)((work / Lq) % M);
    const int64_t n = work / Lq / M;
    const int64_t row = (n * Lq + q) * M + m;
    const int64_t stride = (int64_t)M * kD;
    const int64_t head_off = n * S * stride + m * kD + sub * 4;

    // Register diet (this kernel waits on gathers: time ~ 1 / loads in flight per SIMD, i.e. it is
    // paid in occupancy): the reduced per-sample sums are needed by ONE lane of the row only, so each
    // lane keeps just the samples it will store (s & 7 == sub) instead of all LP x 3 of them.
    constexpr int NM = (LP + 7) / 8;
    float p[LP], ga_m[NM], gx_m[NM], gy_m[NM], p_m[NM];
    row_softmax<PT, LP>(logit + row * LP, p);
#pragma unroll
    for (int i = 0; i < NM; ++i) ga_m[i] = gx_m[i] = gy_m[i] = p_m[i] = 0.f;
    float dot = 0.f;
    const uint32_t row_bytes = (uint32_t)(stride * sizeof(VT));
    const uint32_t lane_off = (uint32_t)(head_off * sizeof(VT));
    const float4 g = load4(grad_out + row * kD + sub * 4);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const Level lv = read_level(shapes, lsi, l, S);
        // 32-bit byte offsets from the (uniform) tensor base: one address register per gather instead
        // of two (the host routes value tensors of 4 GB and more to msda_fused_bwd)
        const int64_t lstart = lv.valid ? lv.start : 0;      // invalid level: rows are -1 -> token 0, selected away
        const uint32_t lvl_off = lane_off + (uint32_t)lstart * row_bytes;
        float *gvl = grad_value + head_off + lstart * stride;
        const TapL *tp = s_tap + rl * LP + l * P;
        float4 v[P][4];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int4 rw = *reinterpret_cast<const int4 *>(tp[u].row);
            const int r4[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // unconditional (clamped) load + select: a load under `if (valid)` gets its own
                // s_waitcnt and the 16 gathers of a row run one after the other
                const float4 x = load4(reinterpret_cast<const VT *>(
                    reinterpret_cast<const char *>(value) + (lvl_off + (uint32_t)max(r4[k], 0) * row_bytes)));
                const bool ok = r4[k] >= 0;
                v[u][k] = make_float4(ok ? x.x : 0.f, ok ? x.y : 0.f, ok ? x.z : 0.f, ok ? x.w : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int s = l * P + u;
            const float a = p[s];
            const float lh = tp[u].lh, lw = tp[u].lw, hh = 1.f - lh, hw = 1.f - lw;
            const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            if (tp[u].far) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tp[u].row[k] >= 0) {
                        float *d = gvl + (int64_t)tp[u].row[k] * stride;
                        const float w = cw[k] * a;
                        atomicAdd(d + 0, w * g.x);
                        atomicAdd(d + 1, w * g.y);
                        atomicAdd(d + 2, w * g.z);
                        atomicAdd(d + 3, w * g.w);
                    }
            }
            // per-lane partial dot products over its 4 channels
            auto dot4 = [&](const float4 &x) { return g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w; };
            const float d0 = dot4(v[u][0]), d1 = dot4(v[u][1]), d2 = dot4(v[u][2]), d3 = dot4(v[u][3]);
            const float val = cw[0] * d0 + cw[1] * d1 + cw[2] * d2 + cw[3] * d3;
            const float gh = hw * (d2 - d0) + lw * (d3 - d1);
            const float gw = hh * (d1 - d0) + lh * (d3 - d2);
            const float gas = sum8(val), gxs = sum8(gw * a), gys = sum8(gh * a);
            dot += a * gas;
            const bool mine = (s & 7) == sub;
            ga_m[s >> 3] = mine ? gas : ga_m[s >> 3];
            gx_m[s >> 3] = mine ? gxs : gx_m[s >> 3];
            gy_m[s >> 3] = mine ? gys : gy_m[s >> 3];
            p_m[s >> 3] = mine ? a : p_m[s >> 3];
        }
    }
    // LP <= 16 samples, 8 lanes: lane `sub` stores samples sub and sub + 8
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int s = i * 8 + sub;
        if (s < LP) {
            d_logit[row * LP + s] = (PT)(p_m[i] * (ga_m[i] - dot));
            store2(d_off + (row * LP + s) * 2, gx_m[i], gy_m[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// grad_value of the NEAR samples without per-sample atomics ("pull" form).
//
// The value maps are cut into tiles; a workgroup owns (batch n, head m, one tile of one level) and
// is given, by the host, the list of CANDIDATE queries whose reference point lies within
// near_radius (+ margin) of the tile - a static function of the reference grid, built once per
// geometry.  It recomputes the candidates' sampling taps, keeps the (sample, corner) pairs that land
// inside its tile, buckets them by pixel with an LDS counting sort (integer LDS atomics on the
// bucket counters only), and then one half-wave per pixel sums  w * grad_out[row]  over the pixel's
// bucket in registers (32 channel lanes) and issues ONE 128-byte atomic add per pixel row.  Every
// near (sample, corner) pair is found by exactly one workgroup (tiles partition the level), far
// samples were scattered by msda_fused_bwd: together they are the whole gradient.
// Replaces 4 x 128 B of memory-side float atomics per sample (the 1.2 TB/s atomic roofline the plain
// backward sits on) with one per pixel row and tile.
// ---------------------------------------------------------------------------------------
struct TileMeta {          // 8 x int32 per tile (host built)
    int level, y0, x0, ny, nx, cand_start, cand_count, pad;
};
constexpr int kMaxTilePx = 256;
constexpr int kGvThreads = 1024;     // 32 half-waves: enough 128-byte row gathers in flight per CU

template <typename PT, int LP>
__device__ __forceinline__ float softmax_weight(const PT *__restrict__ lg, int s) {
    float p[LP];
    row_softmax<PT, LP>(lg, p);
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) r = (i == s) ? p[i] : r;
    return r;
}

template <typename VT, typename PT, int L, int P>
__global__ __launch_bounds__(kGvThreads) void msda_fused_bwd_gv(
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi, const PT *__restrict__ off,
    const PT *__restrict__ logit, const float *__restrict__ ref, int ref_levels,
    const VT *__restrict__ grad_out, const int *__restrict__ tile_meta, const int *__restrict__ cand,
    int ntiles, int64_t S, int M, int64_t Lq, float near_radius, int cap, float *__restrict__ grad_value) {
    constexpr int LP = L * P;
    // dynamic LDS: [records: cap x {row | px << 24, weight bits}] [order: cap x uint16]
    extern __shared__ __attribute__((aligned(16))) int s_rec[];
    unsigned short *s_order = reinterpret_cast<unsigned short *>(s_rec + 2 * (size_t)cap);
    __shared__ int s_cnt[kMaxTilePx], s_start[kMaxTilePx + 1], s_cur[kMaxTilePx];
    __shared__ int s_total;

    const int64_t b = blockIdx.x;
    const int tile = (int)(b % ntiles);
    const int m = (int)((b / ntiles) % M);
    const int64_t n = b / ntiles / M;
    const TileMeta tm = reinterpret_cast<const TileMeta *>(tile_meta)[tile];
    const int l = tm.level;
    const Level lv = read_level(shapes, lsi, l, S);
    const int npx = tm.ny * tm.nx;
    if (!lv.valid || npx <= 0 || npx > kMaxTilePx || tm.cand_count <= 0) return;

    for (int i = threadIdx.x; i < npx; i += kGvThreads) s_cnt[i] = s_cur[i] = 0;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();

    const int nitems = tm.cand_count * P;
    const int lane = threadIdx.x & 63;
    float *gv_level = grad_value + (n * S + lv.start) * (int64_t)M * kD + (int64_t)m * kD;

    // ---- 1: sample the candidates once; the (sample, corner) pairs that land in the tile are
    // appended to the record list with a wave ballot (one LDS atomic per wave and corner, issued by
    // one lane) - the hits are sparse (~20 % of the items), per-lane atomics here cost ~100 us.
    constexpr int IB = 4;
    for (int base = threadIdx.x - lane; base < nitems; base += kGvThreads * IB) {      // wave-uniform trip count
        int64_t qv[IB], rowv[IB];
        float2 ov[IB], rpv[IB];
        bool live[IB];
        int pv[IB];
#pragma unroll
        for (int u = 0; u < IB; ++u) {
            const int item = base + lane + u * kGvThreads;
            live[u] = item < nitems;
            const int ci = live[u] ? item / P : 0;
            pv[u] = live[u] ? item - ci * P : 0;
            qv[u] = cand[tm.cand_start + ci];
            live[u] = live[u] && qv[u] >= 0 && qv[u] < Lq;
            if (!live[u]) qv[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < IB; ++u) {
            rowv[u] = (n * Lq + qv[u]) * M + m;
            ov[u] = load2(off + (rowv[u] * LP + l * P + pv[u]) * 2);
            rpv[u] = *reinterpret_cast<const float2 *>(ref + (qv[u] * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
        }
#pragma unroll
        for (int u = 0; u < IB; ++u) {
            int px[4] = {-1, -1, -1, -1};
            float cw[4] = {0.f, 0.f, 0.f, 0.f};
            bool any = false;
            if (live[u] && fabsf(ov[u].x) <= near_radius && fabsf(ov[u].y) <= near_radius) {   // else far: kernel A
                const Tap<float> t = make_tap<float>(rpv[u].x + ov[u].x / (float)lv.W,
                                                     rpv[u].y + ov[u].y / (float)lv.H, lv.H, lv.W);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    cw[k] = t.cw[k];
                    if (!t.ok[k]) continue;
                    const int y = t.row[k] / lv.W, x = t.row[k] - y * lv.W;
                    const int ry = y - tm.y0, rx = x - tm.x0;
                    if (ry < 0 || ry >= tm.ny || rx < 0 || rx >= tm.nx) continue;
                    px[k] = ry * tm.nx + rx;
                    any = true;
                }
            }
            if (__ballot(any) == 0ull) continue;                    // wave-uniform
            const float a = any ? softmax_weight<PT, LP>(logit + rowv[u] * LP, l * P + pv[u]) : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool hit = px[k] >= 0;
                const unsigned long long mask = __ballot(hit);
                if (mask == 0ull) continue;                         // wave-uniform
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_total, __popcll(mask));
                wbase = __shfl(wbase, 0, 64);
                if (hit) {
                    const int slot = wbase + __popcll(mask & ((1ull << lane) - 1ull));
                    const float w = cw[k] * a;
                    if (slot < cap) {
                        s_rec[2 * slot] = (int)rowv[u] | (px[k] << 24);
                        s_rec[2 * slot + 1] = __float_as_int(w);
                    } else {       // record store full: scatter directly (correct, just slow)
                        const int y = tm.y0 + px[k] / tm.nx, x = tm.x0 + px[k] % tm.nx;
                        float *dst = gv_level + ((int64_t)y * lv.W + x) * (int64_t)M * kD;
                        for (int c = 0; c < kD; ++c) atomicAdd(dst + c, w * (float)grad_out[rowv[u] * kD + c]);
                    }
                }
            }
        }
    }
    __syncthreads();
    const int nrec = min(s_total, cap);
    // ---- 2: bucket sizes, dense (every lane holds a record)
    for (int r = threadIdx.x; r < nrec; r += kGvThreads) atomicAdd(&s_cnt[(unsigned)s_rec[2 * r] >> 24], 1);
    __syncthreads();
    // ---- 3: exclusive scan of the bucket sizes (<= 256 buckets): one wave, 4 buckets per lane
    if (threadIdx.x < 64) {
        int v[4], sum = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ii = lane * 4 + jj;
            v[jj] = ii < npx ? s_cnt[ii] : 0;
            sum += v[jj];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        int run = incl - sum;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ii = lane * 4 + jj;
            if (ii < npx) s_start[ii] = run;
            run += v[jj];
        }
        if (lane == 63) s_start[npx] = incl;
    }
    __syncthreads();
    // ---- 4: order[] = record indices grouped by pixel
    for (int r = threadIdx.x; r < nrec; r += kGvThreads) {
        const int px = (unsigned)s_rec[2 * r] >> 24;
        s_order[s_start[px] + atomicAdd(&s_cur[px], 1)] = (unsigned short)r;
    }
    __syncthreads();
    // ---- 5: one half-wave per pixel row: sum its bucket in registers (16 row gathers in flight per
    // lane - the pass is a pure L2 gather and lives on memory-level parallelism), one 128-byte
    // atomic per row
    const int c = threadIdx.x & 31;
    constexpr int U = 16;
    for (int p = threadIdx.x >> 5; p < npx; p += kGvThreads / 32) {
        const int e0 = s_start[p], e1 = s_start[p + 1];
        float acc = 0.f;
        for (int e = e0; e < e1; e += U) {
            float g[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = s_order[min(e + u, e1 - 1)];
                const int2 en = *reinterpret_cast<const int2 *>(s_rec + 2 * r);
                w[u] = e + u < e1 ? __int_as_float(en.y) : 0.f;
                g[u] = (float)grad_out[(int64_t)(en.x & 0xFFFFFF) * kD + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += w[u] * g[u];
        }
        if (e1 > e0) {
            const int y = tm.y0 + p / tm.nx, x = tm.x0 + p % tm.nx;
            atomicAdd(gv_level + ((int64_t)y * lv.W + x) * (int64_t)M * kD + c, acc);
        }
    }
}

// ---------------------------------------------------------------------------------------
// grad_value of the near samples on the matrix cores ("dense pull", bf16 grad_out rows).
//
// For a tile of <= 64 pixels and a chunk of 64 candidate queries the gradient is the product
//     dV[pixel, channel] = sum_q  W[pixel, q] * G[q, channel],
// W[pixel, q] = sum over the query's samples and corners that land on the pixel of
// (attention weight x bilinear weight): a small DENSE matrix built in LDS by the sampling pass
// itself (sparse LDS adds: ~3 per candidate), G = the candidates' grad_out rows staged
// transposed.  Two waves then run v_mfma_f32_32x32x16_bf16 over it (W split into bf16 hi + lo so
// the weights keep ~16 mantissa bits; G is bf16 as stored).  No hit records, no counting sort and
// no per-record gather of 64-byte grad_out rows through L2 - the two things the sort form spends its
// time on (profiles/r01_msda_pull_backward.txt).  The MFMA work is ~50x the useful flops of
// the sparse form and still only ~14 GFLOP per call.
// Tiles with more than 64 pixels are processed in slabs of 64 (correct, slower): the host builds
// 8x8 tiles for this kernel.
// ---------------------------------------------------------------------------------------
constexpr int kDenseKC = 64;                   // candidates per chunk = 256 threads / 4 points
constexpr int kDenseWS = kDenseKC + 4;         // W row stride (floats)
constexpr int kDenseGS = kDenseKC + 8;         // G^T row stride (bf16): 144 B, 16-byte aligned rows
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 dbf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float df32x16;

// Thread t of a chunk owns candidate t >> 2 and its sampling point t & 3 (P == 4), and the 8-channel
// quarter t & 3 of the candidate's grad_out row: everything it needs hangs off ONE candidate id, so
// a chunk costs one dependent global-load latency (the id of the NEXT chunk is fetched a chunk
// ahead), and at 22 KB of LDS seven workgroups per CU overlap their chunks.
#ifndef VAH_PULL_WAVES
#define VAH_PULL_WAVES 6           // waves per SIMD asked of the compiler for the dense pull kernel: <= 80 registers, six
                                   // workgroups per CU (LDS allows six); 183 -> 172 us per backward call, A/B on one box
#endif
template <typename PT, int L, int P, bool kCompact>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VAH_PULL_WAVES))) void msda_fused_bwd_gv_mfma(
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi, const PT *__restrict__ off,
    const PT *__restrict__ logit, const float *__restrict__ ref, int ref_levels,
    const __bf16 *__restrict__ grad_out, const int *__restrict__ tile_meta, const int *__restrict__ cand,
    int ntiles, int64_t S, int M, int64_t Lq, float near_radius, float *__restrict__ grad_value) {
    static_assert(P == 4, "one thread per (candidate, point): 4 points");
    constexpr int LP = L * P;
    __shared__ __attribute__((aligned(16))) float s_w[64 * kDenseWS];
    __shared__ __attribute__((aligned(16))) __bf16 s_gt[kD * kDenseGS];
    // kCompact: only ~20 % of a tile's candidates put a corner into the tile.  A cheap first pass (one
    // thread per candidate: its 4 offsets of this level + the reference point, 24 of the 100 bytes a
    // candidate costs below) keeps the candidates that do, compacted in candidate order into s_hit; the
    // chunk body then runs over hits only: ~5x fewer grad_out row loads, W builds and MFMA slabs.
    __shared__ int s_hit[kCompact ? 64 + 256 : 1];
    __shared__ int s_cnt[2][4];

    // tile-major order: the host sorts the tiles by candidate count, so the long ones start first
    const int64_t b = blockIdx.x;
    const int NM = (int)(gridDim.x / ntiles);
    const int tile = (int)(b / NM);
    const int m = (int)(b % NM) % M;
    const int64_t n = (b % NM) / M;
    const TileMeta tm = reinterpret_cast<const TileMeta *>(tile_meta)[tile];
    const int l = tm.level;
    const Level lv = read_level(shapes, lsi, l, S);
    const int npx = tm.ny * tm.nx;
    if (!lv.valid || npx <= 0 || tm.cand_count <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    const int ci = tid >> 2, p = tid & 3;
    float *gv_level = grad_value + (n * S + lv.start) * (int64_t)M * kD + (int64_t)m * kD;

    for (int px0 = 0; px0 < npx; px0 += 64) {
        df32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        int q_next = (!kCompact && ci < tm.cand_count) ? cand[tm.cand_start + ci] : -1;
        int qa_next = (kCompact && tid < tm.cand_count) ? cand[tm.cand_start + tid] : -1;      // one pass ahead
        int c0 = 0, nhit = 0, pass = 0;                 // uniform over the workgroup
        for (;;) {
            int q;
            if constexpr (kCompact) {
                // ---- first pass: top the hit list up to one chunk (64) or the end of the candidates
                while (nhit < kDenseKC && c0 < tm.cand_count) {
                    int qa = qa_next;
                    if (qa < 0 || qa >= Lq) qa = -1;
                    {
                        const int nx = c0 + 256 + tid;
                        qa_next = nx < tm.cand_count ? cand[tm.cand_start + nx] : -1;
                    }
                    bool hit = false;
                    if (qa >= 0) {
                        const int64_t rowa = (n * Lq + qa) * M + m;
                        const float2 rpa = *reinterpret_cast<const float2 *>(
                            ref + ((int64_t)qa * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
                        float2 oa[P];
#pragma unroll
                        for (int u = 0; u < P; ++u) oa[u] = load2(off + (rowa * LP + l * P + u) * 2);
#pragma unroll
                        for (int u = 0; u < P; ++u) {          // the chunk body's own arithmetic: an exact filter
                            if (!(fabsf(oa[u].x) <= near_radius && fabsf(oa[u].y) <= near_radius)) continue;
                            const float lxa = rpa.x + oa[u].x / (float)lv.W, lya = rpa.y + oa[u].y / (float)lv.H;
                            const float h_im = lya * (float)lv.H - 0.5f, w_im = lxa * (float)lv.W - 0.5f;
                            if (!(h_im > -1.f && w_im > -1.f && h_im < (float)lv.H && w_im < (float)lv.W)) continue;
                            const int ry0 = (int)floorf(h_im) - tm.y0, rx0 = (int)floorf(w_im) - tm.x0;
                            const bool yin = (ry0 >= 0 && ry0 < tm.ny) || (ry0 + 1 >= 0 && ry0 + 1 < tm.ny);
                            const bool xin = (rx0 >= 0 && rx0 < tm.nx) || (rx0 + 1 >= 0 && rx0 + 1 < tm.nx);
                            hit = hit || (yin && xin);
                        }
                    }
                    const unsigned long long bal = __ballot(hit);
                    if (lane == 0) s_cnt[pass & 1][wv] = __popcll(bal);
                    __syncthreads();
                    int base = nhit, total = 0;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const int cw = s_cnt[pass & 1][w2];
                        base += w2 < wv ? cw : 0;
                        total += cw;
                    }
                    if (hit) s_hit[base + __popcll(bal & ((1ull << lane) - 1ull))] = qa;
                    nhit += total;
                    c0 += 256;
                    ++pass;
                }
                if (nhit == 0) break;                    // list drained and no candidates left
                __syncthreads();                         // the hits are visible
                q = ci < min(nhit, kDenseKC) ? s_hit[ci] : -1;
            } else {
                if (c0 >= tm.cand_count) break;
                q = q_next;
                if (q < 0 || q >= Lq) q = -1;
                const int nxt = c0 + kDenseKC + ci;
                q_next = nxt < tm.cand_count ? cand[tm.cand_start + nxt] : -1;
                c0 += kDenseKC;
            }
            // ---- issue every load of this thread's (candidate, point) at once
            const int64_t row = (n * Lq + max(q, 0)) * M + m;
            const dbf16x8 gq = *reinterpret_cast<const dbf16x8 *>(grad_out + row * kD + p * 8);
            const float2 o = load2(off + (row * LP + l * P + p) * 2);
            const float2 rp = *reinterpret_cast<const float2 *>(ref + ((int64_t)max(q, 0) * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
            float pr[LP];                          // raw logits now, softmax only if a corner lands in the tile
#pragma unroll
            for (int s = 0; s < LP; ++s) pr[s] = (float)logit[row * LP + s];
            // ---- zero W, stage the grad_out quarter-row transposed
            for (int i = tid; i < 64 * kDenseWS / 4; i += 256)
                reinterpret_cast<float4 *>(s_w)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_gt[(p * 8 + e) * kDenseGS + ci] = q >= 0 ? gq[e] : (__bf16)0.f;
            __syncthreads();
            // ---- the point's in-tile corners -> W[pixel][candidate]; same location arithmetic as make_tap
            if (q >= 0 && fabsf(o.x) <= near_radius && fabsf(o.y) <= near_radius) {      // else far: kernel A
                const float lx = rp.x + o.x / (float)lv.W, ly = rp.y + o.y / (float)lv.H;
                const float h_im = ly * (float)lv.H - 0.5f, w_im = lx * (float)lv.W - 0.5f;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)lv.H && w_im < (float)lv.W) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int ry0 = (int)hf - tm.y0, rx0 = (int)wf - tm.x0;       // the tile lies inside the map
                    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                    const bool y_in[2] = {ry0 >= 0 && ry0 < tm.ny, ry0 + 1 >= 0 && ry0 + 1 < tm.ny};
                    const bool x_in[2] = {rx0 >= 0 && rx0 < tm.nx, rx0 + 1 >= 0 && rx0 + 1 < tm.nx};
                    if ((y_in[0] || y_in[1]) && (x_in[0] || x_in[1])) {
                        float mx = -INFINITY, sum = 0.f, mine = 0.f;
#pragma unroll
                        for (int s = 0; s < LP; ++s) mx = fmaxf(mx, pr[s]);
#pragma unroll
                        for (int s = 0; s < LP; ++s) {
                            const float e = __expf(pr[s] - mx);
                            sum += e;
                            mine = (s == l * P + p) ? e : mine;
                        }
                        const float a = mine * (1.f / sum);
                        const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (!(y_in[k >> 1] && x_in[k & 1])) continue;
                            const int pp = (ry0 + (k >> 1)) * tm.nx + rx0 + (k & 1) - px0;
                            if (pp < 0 || pp >= 64) continue;
                            atomicAdd(&s_w[pp * kDenseWS + ci], cw[k] * a);
                        }
                    }
                }
            }
            __syncthreads();
            // ---- dV[64 px, 32 ch] += W[64, KC] G[KC, 32]: wave w owns pixels 32(w&1) .. +31 and the
            // candidates 32(w>>1) .. +31 of the chunk (partial sums of the two halves meet in the atomics)
            {
                const float *wrow = s_w + ((wv & 1) * 32 + r) * kDenseWS + 8 * h + 32 * (wv >> 1);
                const __bf16 *grow = s_gt + r * kDenseGS + 8 * h + 32 * (wv >> 1);
#pragma unroll
                for (int ks = 0; ks < kDenseKC / 32; ++ks) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(wrow + 16 * ks);
                    const float4 w1 = *reinterpret_cast<const float4 *>(wrow + 16 * ks + 4);
                    const float wf[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                    dbf16x8 ahi, alo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        ahi[e] = (__bf16)wf[e];
                        alo[e] = (__bf16)(wf[e] - (float)ahi[e]);
                    }
                    const dbf16x8 bg = *reinterpret_cast<const dbf16x8 *>(grow + 16 * ks);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bg, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bg, acc, 0, 0, 0);
                }
            }
            __syncthreads();
            if constexpr (kCompact) {                    // drop the processed chunk from the hit list
                const int rest = max(nhit - kDenseKC, 0);          // <= 255
                const int keep = tid < rest ? s_hit[kDenseKC + tid] : 0;
                __syncthreads();
                if (tid < rest) s_hit[tid] = keep;
                nhit = rest;
                // (the next top-up pass or chunk starts with a barrier of its own before s_hit is read)
            }
        }
        // ---- one 128-byte atomic per pixel row and wave: lane = channel r, registers = pixel rows
        {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int pp = px0 + (wv & 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (pp < npx) {
                    const int y = tm.y0 + pp / tm.nx, x = tm.x0 + pp % tm.nx;
                    atomicAdd(gv_level + ((int64_t)y * lv.W + x) * (int64_t)M * kD + r, acc[i]);
                }
            }
        }
    }
}

struct FusedArgs {
    const void *value, *off, *logit, *grad_out;
    const int64_t *shapes, *lsi;
    const float *ref;
    int ref_levels;
    int64_t N, S, M, L, Lq, P;
    void *out;
    float *grad_value;
    void *d_off, *d_logit;
    const int *tile_meta = nullptr, *cand = nullptr;    // optional pull schedule
    int64_t ntiles = 0, cap = 0;
    float near_radius = -1.f;
    bool taps_only = false;          // d(offsets), d(logits) only: grad_value belongs to the tile pass (msda_tile.hip)
    hipStream_t st;
};

template <typename VT, typename PT, int L, int P>
int launch_fwd(const FusedArgs &a) {
    const int64_t rows = a.N * a.Lq * a.M;
    const int64_t nblocks = (rows + (kBlock / 8) - 1) / (kBlock / 8);
    const int64_t grid = (nblocks + 7) / 8 * 8;
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda fused forward: grid too large");
    hipLaunchKernelGGL((msda_fused_fwd<VT, PT, L, P>), dim3((unsigned)grid), dim3(kBlock), 0, a.st,
                       (const VT *)a.value, a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref,
                       a.ref_levels, a.S, (int)a.M, a.Lq, rows, nblocks, (VT *)a.out);
    return check_launch("msda fused forward launch");
}

template <typename VT, typename PT, int L, int P>
int launch_bwd(const FusedArgs &a) {
    const int64_t rows = a.N * a.Lq * a.M;
    const int64_t nblocks = (rows + (kBlock / kD) - 1) / (kBlock / kD);
    const int64_t grid = (nblocks + 7) / 8 * 8;
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda fused backward: grid too large");
    const bool wide = a.N * a.S * a.M * kD * (int64_t)sizeof(VT) >= ((int64_t)1 << 32);   // vec4 kernel: 32-bit offsets
    if ((!a.tile_meta && !a.taps_only) || wide) {
        hipLaunchKernelGGL((msda_fused_bwd<VT, PT, L, P>), dim3((unsigned)grid), dim3(kBlock), 0, a.st,
                           (const VT *)a.value, a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref,
                           a.ref_levels, (const VT *)a.grad_out, a.S, (int)a.M, a.Lq, rows, nblocks,
                           a.taps_only ? kNoScatter : -1.f, a.grad_value, (PT *)a.d_off, (PT *)a.d_logit);
        return check_launch("msda fused backward launch");
    }
    {
        const int64_t nb8 = (rows + (kBlock / 8) - 1) / (kBlock / 8);
        const int64_t grid8 = (nb8 + 7) / 8 * 8;
        hipLaunchKernelGGL((msda_fused_bwd_vec4<VT, PT, L, P>), dim3((unsigned)grid8), dim3(kBlock), 0, a.st,
                           (const VT *)a.value, a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref,
                           a.ref_levels, (const VT *)a.grad_out, a.S, (int)a.M, a.Lq, rows, nb8,
                           a.taps_only ? kNoScatter : a.near_radius, a.grad_value, (PT *)a.d_off, (PT *)a.d_logit);
        if (int rc = check_launch("msda fused backward (split) launch")) return rc;
    }
    if (a.taps_only) return VAH_OK;
    // pull pass for the near samples
    const int64_t gblocks = a.N * a.M * a.ntiles;
    if (gblocks >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda fused backward: tile grid too large");
    if (a.cap == 0) {                            // dense pull on the matrix cores (bf16 grad_out rows only)
        if constexpr (std::is_same<VT, __bf16>::value) {
            static const bool compact = [] {
                const char *e = getenv("VAH_MSDA_PULL_COMPACT");
                return !(e && e[0] == '0');
            }();
            if (compact)
                hipLaunchKernelGGL((msda_fused_bwd_gv_mfma<PT, L, P, true>), dim3((unsigned)gblocks), dim3(256), 0, a.st,
                                   a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref, a.ref_levels,
                                   (const __bf16 *)a.grad_out, a.tile_meta, a.cand, (int)a.ntiles, a.S, (int)a.M, a.Lq,
                                   a.near_radius, a.grad_value);
            else
                hipLaunchKernelGGL((msda_fused_bwd_gv_mfma<PT, L, P, false>), dim3((unsigned)gblocks), dim3(256), 0, a.st,
                                   a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref, a.ref_levels,
                                   (const __bf16 *)a.grad_out, a.tile_meta, a.cand, (int)a.ntiles, a.S, (int)a.M, a.Lq,
                                   a.near_radius, a.grad_value);
            return check_launch("msda fused backward (dense pull) launch");
        } else {
            return fail(VAH_E_UNSUPPORTED, "msda fused backward: the dense pull (cap_entries = 0) needs bf16 values");
        }
    }
    const size_t smem = (size_t)a.cap * 10;      // records (8 B) + order (2 B)
    if (int rc = allow_dynamic_lds((const void *)msda_fused_bwd_gv<VT, PT, L, P>, 150 * 1024, "msda fused backward"))
        return rc;
    hipLaunchKernelGGL((msda_fused_bwd_gv<VT, PT, L, P>), dim3((unsigned)gblocks), dim3(kGvThreads), smem, a.st,
                       a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref, a.ref_levels,
                       (const VT *)a.grad_out, a.tile_meta, a.cand, (int)a.ntiles, a.S, (int)a.M, a.Lq,
                       a.near_radius, (int)a.cap, a.grad_value);
    return check_launch("msda fused backward (grad_value pull) launch");
}

template <bool BWD, typename VT, typename PT>
int dispatch_lp(const FusedArgs &a) {
#define VAH_CASE(LL, PP)                                                        \
    if (a.L == LL && a.P == PP) return BWD ? launch_bwd<VT, PT, LL, PP>(a) : launch_fwd<VT, PT, LL, PP>(a)
    VAH_CASE(1, 4);
    VAH_CASE(3, 4);
    VAH_CASE(4, 4);
#undef VAH_CASE
    return fail(VAH_E_UNSUPPORTED, "msda fused: (L, P) = (%lld, %lld) not instantiated", (long long)a.L, (long long)a.P);
}

template <bool BWD>
int dispatch(const FusedArgs &a, int value_dtype, int param_dtype) {
    if (value_dtype == 0 && param_dtype == 0) return dispatch_lp<BWD, float, float>(a);
    if (value_dtype == 1 && param_dtype == 1) return dispatch_lp<BWD, __bf16, __bf16>(a);
    if (value_dtype == 1 && param_dtype == 0) return dispatch_lp<BWD, __bf16, float>(a);
    if (value_dtype == 0 && param_dtype == 1) return dispatch_lp<BWD, float, __bf16>(a);
    return fail(VAH_E_UNSUPPORTED, "msda fused: dtype codes must be 0 (f32) or 1 (bf16)");
}

int check_common(const char *fn, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                 int64_t ref_levels) {
    if (N < 0 || S < 1 || M < 1 || L < 1 || Lq < 0 || P < 1 || M * D >= (1LL << 31))
        return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (D != kD) return fail(VAH_E_UNSUPPORTED, "%s: needs D == 32", fn);
    if (ref_levels != 1 && ref_levels != L) return fail(VAH_E_SHAPE, "%s: ref_levels must be 1 or L", fn);
    return VAH_OK;
}

}  // namespace

int msda_fused_grad_taps(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi, const void *offsets,
                         const void *logits, int param_dtype, const float *ref, int64_t ref_levels, const void *grad_out,
                         int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq, int64_t P, void *d_offsets, void *d_logits,
                         hipStream_t st) {
    FusedArgs a{};
    a.value = value, a.off = offsets, a.logit = logits, a.shapes = shapes, a.lsi = lsi, a.ref = ref;
    a.ref_levels = (int)ref_levels, a.N = N, a.S = S, a.M = M, a.L = L, a.Lq = Lq, a.P = P;
    a.grad_out = grad_out, a.d_off = d_offsets, a.d_logit = d_logits;
    a.grad_value = nullptr;          // never dereferenced: nothing scatters in this mode
    a.taps_only = true;
    a.st = st;
    return dispatch<true>(a, value_dtype, param_dtype);
}

}  // namespace vah

extern "C" {

int vah_msda_fused_supported(int64_t D, int64_t L, int64_t P) {
    return D == 32 && P == 4 && (L == 1 || L == 3 || L == 4);
}

int vah_msda_fused_forward(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                           const void *offsets, const void *logits, int param_dtype, const float *ref,
                           int64_t ref_levels, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L,
                           int64_t Lq, int64_t P, void *out, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_fused_forward";
    if (int rc = check_common(fn, N, S, M, D, L, Lq, P, ref_levels)) return rc;
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !offsets || !logits || !ref || !out) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)value | (uintptr_t)out | (uintptr_t)offsets | (uintptr_t)ref) % 8) return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    FusedArgs a{};
    a.value = value, a.off = offsets, a.logit = logits, a.shapes = shapes, a.lsi = lsi, a.ref = ref;
    a.ref_levels = (int)ref_levels, a.N = N, a.S = S, a.M = M, a.L = L, a.Lq = Lq, a.P = P, a.out = out;
    a.st = (hipStream_t)stream;
    // algorithmic bytes the launch really moves: value / out in the value dtype, offsets (2) + logits (1)
    // per sample in the parameter dtype; the op's fp32 definition is reported
    // beside it as def_bytes
    const int64_t vs = value_dtype == 1 ? 2 : 4, ps = param_dtype == 1 ? 2 : 4;
    LaunchScope scope("msda_fused_fwd", vs * (N * S * M * D + N * Lq * M * D) + ps * 3 * N * Lq * M * L * P, a.st,
                      4 * (N * S * M * D + 3 * N * Lq * M * L * P + N * Lq * M * D));
    return dispatch<false>(a, value_dtype, param_dtype);
}

// grad_value: fp32 (N,S,M,D), zero on entry.  grad_out has the value dtype; d_offsets / d_logits the
// parameter dtype.
int vah_msda_fused_backward(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                            const void *offsets, const void *logits, int param_dtype, const float *ref,
                            int64_t ref_levels, const void *grad_out, int64_t N, int64_t S, int64_t M,
                            int64_t D, int64_t L, int64_t Lq, int64_t P, float *grad_value,
                            void *d_offsets, void *d_logits, const int32_t *tile_meta,
                            const int32_t *cand, int64_t ntiles, float near_radius, int64_t cap_entries,
                            void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_fused_backward";
    if (tile_meta && (!cand || ntiles < 1 || near_radius < 0.f ||
                      (cap_entries != 0 && (cap_entries < 64 || cap_entries * 10 > 150 * 1024 || cap_entries > 65535 ||
                                            (cap_entries & 1)))))
        return fail(VAH_E_SHAPE, "%s: bad pull schedule", fn);
    if (tile_meta && cap_entries != 0 && N * Lq * M >= (1 << 24)) tile_meta = nullptr;     // packed row index has 24 bits: plain path
    if (tile_meta && Lq >= ((int64_t)1 << 31)) tile_meta = nullptr;
    if (int rc = check_common(fn, N, S, M, D, L, Lq, P, ref_levels)) return rc;
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !offsets || !logits || !ref || !grad_out || !grad_value || !d_offsets || !d_logits)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    FusedArgs a{};
    a.value = value, a.off = offsets, a.logit = logits, a.shapes = shapes, a.lsi = lsi, a.ref = ref;
    a.ref_levels = (int)ref_levels, a.N = N, a.S = S, a.M = M, a.L = L, a.Lq = Lq, a.P = P;
    a.grad_out = grad_out, a.grad_value = grad_value, a.d_off = d_offsets, a.d_logit = d_logits;
    a.tile_meta = tile_meta, a.cand = cand, a.ntiles = ntiles, a.cap = cap_entries, a.near_radius = near_radius;
    a.st = (hipStream_t)stream;
    // moved bytes: value + grad_out read in the value dtype, offsets / logits read and their gradients
    // written in the parameter dtype, grad_value written in fp32 (its zero-fill is not counted)
    const int64_t vs = value_dtype == 1 ? 2 : 4, ps = param_dtype == 1 ? 2 : 4;
    LaunchScope scope("msda_fused_bwd", vs * (N * S * M * D + N * Lq * M * D) + 4 * N * S * M * D + ps * 6 * N * Lq * M * L * P,
                      a.st, 4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D));
    return dispatch<true>(a, value_dtype, param_dtype);
}

}  // extern "C"
