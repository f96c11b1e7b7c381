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
    const PT *__restrict__ off, const PT *__restrict__ logit, int64_t os, int64_t ls, const float *__restrict__ ref,
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
            const float2 o = load2(off + rw * os + sidx * 2);
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
    row_softmax<PT, LP>(logit + row * ls, p);
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
// near_radius == kNoScatter: grad_value is left to the tile pass (msda_tile.hip) and nothing is scattered here;
// otherwise every sample's four corners are scattered with float atomics.
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
            scatter[u] = near_radius != kNoScatter;
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

// Gather-only variant of the backward above: d(offsets), d(logits), no grad_value (the tile pass of msda_tile.hip
// owns it).  8 lanes x 4 channels per row (16-byte corner loads as in the forward: 4x fewer gather instructions
// than one channel per lane).
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
    int pad[2];
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
        tl.pad[0] = tl.pad[1] = 0;
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

struct FusedArgs {
    const void *value, *off, *logit, *grad_out;
    const int64_t *shapes, *lsi;
    const float *ref;
    int ref_levels;
    int64_t N, S, M, L, Lq, P;
    void *out;
    float *grad_value;
    void *d_off, *d_logit;
    int64_t os = 0, ls = 0;          // forward only: elements between the offsets / logits of consecutive rows
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
                       (const VT *)a.value, a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.os ? a.os : L * P * 2,
                       a.ls ? a.ls : L * P, a.ref, a.ref_levels, a.S, (int)a.M, a.Lq, rows, nblocks, (VT *)a.out);
    return check_launch("msda fused forward launch");
}

template <typename VT, typename PT, int L, int P>
int launch_bwd(const FusedArgs &a) {
    const int64_t rows = a.N * a.Lq * a.M;
    const int64_t nblocks = (rows + (kBlock / kD) - 1) / (kBlock / kD);
    const int64_t grid = (nblocks + 7) / 8 * 8;
    if (grid >= ((int64_t)1 << 31)) return fail(VAH_E_SHAPE, "msda fused backward: grid too large");
    const bool wide = a.N * a.S * a.M * kD * (int64_t)sizeof(VT) >= ((int64_t)1 << 32);   // vec4 kernel: 32-bit offsets
    if (!a.taps_only || wide) {
        // one lane per channel; scatters grad_value with atomics unless the tile pass owns it (taps_only)
        hipLaunchKernelGGL((msda_fused_bwd<VT, PT, L, P>), dim3((unsigned)grid), dim3(kBlock), 0, a.st,
                           (const VT *)a.value, a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref,
                           a.ref_levels, (const VT *)a.grad_out, a.S, (int)a.M, a.Lq, rows, nblocks,
                           a.taps_only ? kNoScatter : -1.f, a.grad_value, (PT *)a.d_off, (PT *)a.d_logit);
        return check_launch("msda fused backward launch");
    }
    const int64_t nb8 = (rows + (kBlock / 8) - 1) / (kBlock / 8);
    const int64_t grid8 = (nb8 + 7) / 8 * 8;
    hipLaunchKernelGGL((msda_fused_bwd_vec4<VT, PT, L, P>), dim3((unsigned)grid8), dim3(kBlock), 0, a.st,
                       (const VT *)a.value, a.shapes, a.lsi, (const PT *)a.off, (const PT *)a.logit, a.ref,
                       a.ref_levels, (const VT *)a.grad_out, a.S, (int)a.M, a.Lq, rows, nb8, kNoScatter,
                       a.grad_value, (PT *)a.d_off, (PT *)a.d_logit);
    return check_launch("msda fused backward (gather) launch");
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
                           const void *offsets, const void *logits, int param_dtype, int64_t offsets_stride,
                           int64_t logits_stride, const float *ref,
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
    a.os = offsets_stride, a.ls = logits_stride;
    if (offsets_stride < 0 || logits_stride < 0 || ((offsets_stride * (param_dtype == 1 ? 2 : 4)) % 8) ||
        ((logits_stride * (param_dtype == 1 ? 2 : 4)) % (param_dtype == 1 ? 2 : 4)))
        return fail(VAH_E_ALIGN, "%s: bad strides", fn);
    a.st = (hipStream_t)stream;
    // algorithmic bytes the launch really moves: value / out in the value dtype, offsets (2) + logits (1)
    // per sample in the parameter dtype; the op's fp32 definition is reported
    // beside it as def_bytes
    const int64_t vs = value_dtype == 1 ? 2 : 4, ps = param_dtype == 1 ? 2 : 4;
    LaunchScope scope("msda_fused_fwd", vs * (N * S * M * D + N * Lq * M * D) + ps * 3 * N * Lq * M * L * P, a.st,
                      4 * (N * S * M * D + 3 * N * Lq * M * L * P + N * Lq * M * D));
    return dispatch<false>(a, value_dtype, param_dtype);
}

// grad_value: fp32 (N,S,M,D), zero on entry (float atomics).  grad_out has the value dtype; d_offsets / d_logits the
// parameter dtype.  This is the fallback of vah_msda_fused_backward_tiled (csrc/msda_tile.hip) for the shapes the tile
// pass does not take.
int vah_msda_fused_backward(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                            const void *offsets, const void *logits, int param_dtype, const float *ref,
                            int64_t ref_levels, const void *grad_out, int64_t N, int64_t S, int64_t M,
                            int64_t D, int64_t L, int64_t Lq, int64_t P, float *grad_value,
                            void *d_offsets, void *d_logits, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_fused_backward";
    if (int rc = check_common(fn, N, S, M, D, L, Lq, P, ref_levels)) return rc;
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !offsets || !logits || !ref || !grad_out || !grad_value || !d_offsets || !d_logits)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    FusedArgs a{};
    a.value = value, a.off = offsets, a.logit = logits, a.shapes = shapes, a.lsi = lsi, a.ref = ref;
    a.ref_levels = (int)ref_levels, a.N = N, a.S = S, a.M = M, a.L = L, a.Lq = Lq, a.P = P;
    a.grad_out = grad_out, a.grad_value = grad_value, a.d_off = d_offsets, a.d_logit = d_logits;
    a.st = (hipStream_t)stream;
    // moved bytes: value + grad_out read in the value dtype, offsets / logits read and their gradients
    // written in the parameter dtype, grad_value written in fp32 (its zero-fill is not counted)
    const int64_t vs = value_dtype == 1 ? 2 : 4, ps = param_dtype == 1 ? 2 : 4;
    LaunchScope scope("msda_fused_bwd", vs * (N * S * M * D + N * Lq * M * D) + 4 * N * S * M * D + ps * 6 * N * Lq * M * L * P,
                      a.st, 4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D));
    return dispatch<true>(a, value_dtype, param_dtype);
}

}  // extern "C"
