// Output tail of the adapter backbone:  f = BatchNorm(a + b + upsample_s(x)).
//
// Reference: vit_adapter.py:106-127 (seg) / :101-120 (det)
//     c1 = up(c2) + c1;  c1 = c1 + F.interpolate(x1, scale_factor=4, mode='bilinear');  f1 = norm1(c1)
//     c2 = c2 + F.interpolate(x2, scale_factor=2, ...);  f2 = norm2(c2);  c3 = c3 + x3;  f3 = norm3(c3)
// At 1024^2 the stride-4 map is 2 x 768 x 256 x 256 = 100 M elements; as separate ops (interpolate,
// two adds, BN statistics, BN normalise) it is written or read ~10 times in fp32 (3.8 GB forward).
// Here the sum is never materialised: the statistics pass and the normalise pass both recompute
// a + b + bilinear(x) from the (bf16) operands, so the forward moves 2 reads of the operands plus
// one fp32 write, and the backward likewise.  HBM-bound; no LDS except the adjoint of the upsample.
//
// Layout: NCHW planes.  a: bf16 or fp32 (N, C, H, W); b: optional, bf16 or fp32, same shape;
// x: optional fp32 (N, C, H / s, W / s), s in {1, 2, 4, 8} (s = 1: plain add).  W % 4 == 0.
// Bilinear taps follow torch's upsample_bilinear2d with align_corners = False and an explicit
// scale factor:  src = (dst + 0.5) / s - 0.5, clamped at 0;  i1 = min(i0 + 1, n - 1).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/vitadapter_hip.h"
#include "common.h"

namespace vah {
namespace {

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;

constexpr int kTilePx = 8192;           // hi-res pixels of one plane per workgroup
constexpr int kTailParts = 512;         // partial rows of the channel sums (finalised below)

struct Operands {
    const void *a, *b;
    const float *x;
    int a_bf16, b_bf16, scale;
    int C, H, W, Hl, Wl;
    int rows_per_block, chunks;         // chunks of rows per plane
    int relu, y_bf16, dy_bf16;          // ReLU fused behind the normalisation; dtypes of y and dy
    const float *shift;                 // optional per-channel constant added to the sum (conv biases)
    unsigned quad_magic;                // floor(2^32 / (W / 4)) + 1: item -> (row, quad) by one v_mul_hi (items < 2^16)
};

struct Tap {
    int i0, i1;
    float w1;
};

// scale = 2^k (host-checked): src = (2d + 1 - s) / 2s exactly, so floor and fraction are integer ops;
// same values as the float form  src = (d + 0.5) / s - 0.5, clamped at 0  (both are exact in fp32).
__device__ __forceinline__ Tap tap_of(int d, int n_lo, int s) {
    const int sh1 = 32 - __builtin_clz((unsigned)s);            // log2(2s)
    const int n = max(2 * d + 1 - s, 0);
    Tap t;
    t.i0 = min(n >> sh1, n_lo - 1);
    t.i1 = min(t.i0 + 1, n_lo - 1);
    t.w1 = (float)(n & (2 * s - 1)) * (0.5f / (float)s);
    return t;
}

__device__ __forceinline__ float4 load4(const void *p, int64_t idx, int is_bf16) {
    if (is_bf16) {
        const bf16x4 v = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const __bf16 *>(p) + idx);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + idx);
}

// Raw 4-element load (4 fp32 or 4 bf16, dtype uniform) and its decode, kept apart so that a kernel can
// request all of its operands before it touches the first: with the conversion next to the load (load4)
// every operand sat in its own uniform branch with its own s_waitcnt vmcnt(0).
struct Raw4 {
    uint32_t w0, w1, w2, w3;
};

__device__ __forceinline__ Raw4 load_raw(const void *p, int64_t idx, int is_bf16) {
    Raw4 r;
    r.w0 = r.w1 = r.w2 = r.w3 = 0u;
    if (is_bf16) {
        const uint2 v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __bf16 *>(p) + idx);
        r.w0 = v.x;
        r.w1 = v.y;
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(p) + idx);
        r.w0 = v.x;
        r.w1 = v.y;
        r.w2 = v.z;
        r.w3 = v.w;
    }
    return r;
}

__device__ __forceinline__ float4 decode(const Raw4 &r, int is_bf16) {
    const float4 h = make_float4(__uint_as_float(r.w0 << 16), __uint_as_float(r.w0 & 0xffff0000u),
                                 __uint_as_float(r.w1 << 16), __uint_as_float(r.w1 & 0xffff0000u));
    const float4 f = make_float4(__uint_as_float(r.w0), __uint_as_float(r.w1), __uint_as_float(r.w2),
                                 __uint_as_float(r.w3));
    return is_bf16 ? h : f;
}

__device__ __forceinline__ void store4(void *p, int64_t idx, int is_bf16, float4 v) {
    if (is_bf16) {
        bf16x4 o;
        o[0] = (__bf16)v.x;
        o[1] = (__bf16)v.y;
        o[2] = (__bf16)v.z;
        o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(p) + idx) = o;
    } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + idx) = v;
    }
}

// t = a + b + up(x) for 4 consecutive pixels (row y, columns x4 .. x4 + 3) of plane `plane`.
// Every load (a, b, the 8 low-res values, and the caller's dy when `dy` is given) is requested before the
// first is decoded.
__device__ __forceinline__ float4 sum4(const Operands &o, int64_t plane, int y, int x4, const void *dy = nullptr,
                                       Raw4 *dy_raw = nullptr) {
    const int64_t idx = (plane * o.H + y) * o.W + x4;
    const Raw4 ra = load_raw(o.a, idx, o.a_bf16);
    Raw4 rb;
    rb.w0 = rb.w1 = rb.w2 = rb.w3 = 0u;
    if (o.b) rb = load_raw(o.b, idx, o.b_bf16);
    if (dy) *dy_raw = load_raw(dy, idx, o.dy_bf16);
    float sft = 0.f;
    if (o.shift) sft = o.shift[plane % o.C];
    float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);          // scale 1: the low-res row itself
    float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
    Tap ty;
    ty.i0 = ty.i1 = 0;
    ty.w1 = 0.f;
    int nx0 = 0, bcol = 0;
    const int sc = o.scale, sh1 = 32 - __builtin_clz((unsigned)sc), mask = 2 * sc - 1;
    if (o.x) {
        const float *xp = o.x + plane * o.Hl * o.Wl;
        if (sc == 1) {
            xs = *reinterpret_cast<const float4 *>(xp + (int64_t)y * o.Wl + x4);
        } else {
            // 4 consecutive pixels starting at a multiple of 4 read at most 4 consecutive low-res columns
            // (b .. b + 3, b = floor of the first pixel's source position): 8 loads instead of 16, the
            // pixel's pair picked with selects.  Index clamping at the borders gives what the clamped
            // source position gives: both taps on the same column, r + w * (r - r) = r.
            ty = tap_of(y, o.Hl, sc);
            const float *r0 = xp + (int64_t)ty.i0 * o.Wl, *r1 = xp + (int64_t)ty.i1 * o.Wl;
            nx0 = 2 * x4 + 1 - sc;
            bcol = nx0 >> sh1;                               // arithmetic shift: floor, >= -1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cj = min(max(bcol + j, 0), o.Wl - 1);
                c0[j] = r0[cj];
                c1[j] = r1[cj];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);                       // loads above, arithmetic below
    float4 t = decode(ra, o.a_bf16);
    const float4 vb = decode(rb, o.b_bf16);
    t.x += sft + vb.x;
    t.y += sft + vb.y;
    t.z += sft + vb.z;
    t.w += sft + vb.w;
    if (o.x) {
        if (sc == 1) {
            t.x += xs.x;
            t.y += xs.y;
            t.z += xs.z;
            t.w += xs.w;
        } else {
            float u[4];
            // The four pixels start at a multiple of 4, so for s = 2 and s = 4 which low-res column pair a pixel reads and
            // with which weight is the same for every quad: constants instead of shifts, masks and 16 selects per quad
            // (this pass is VALU-bound: the statistics pass over the stride-4 level took as long as the normalise pass,
            // which also WRITES 403 MB).  Other scales take the general form.
            if (sc == 4) {                  // columns b, b, b + 1, b + 1 of (c[0..2]); weights 5/8, 7/8, 1/8, 3/8
                const float t0 = c0[0] + 0.625f * (c0[1] - c0[0]), b0 = c1[0] + 0.625f * (c1[1] - c1[0]);
                const float t1 = c0[0] + 0.875f * (c0[1] - c0[0]), b1 = c1[0] + 0.875f * (c1[1] - c1[0]);
                const float t2 = c0[1] + 0.125f * (c0[2] - c0[1]), b2 = c1[1] + 0.125f * (c1[2] - c1[1]);
                const float t3 = c0[1] + 0.375f * (c0[2] - c0[1]), b3 = c1[1] + 0.375f * (c1[2] - c1[1]);
                u[0] = t0 + ty.w1 * (b0 - t0);
                u[1] = t1 + ty.w1 * (b1 - t1);
                u[2] = t2 + ty.w1 * (b2 - t2);
                u[3] = t3 + ty.w1 * (b3 - t3);
            } else if (sc == 2) {           // columns b, b + 1, b + 1, b + 2; weights 3/4, 1/4, 3/4, 1/4
                const float t0 = c0[0] + 0.75f * (c0[1] - c0[0]), b0 = c1[0] + 0.75f * (c1[1] - c1[0]);
                const float t1 = c0[1] + 0.25f * (c0[2] - c0[1]), b1 = c1[1] + 0.25f * (c1[2] - c1[1]);
                const float t2 = c0[1] + 0.75f * (c0[2] - c0[1]), b2 = c1[1] + 0.75f * (c1[2] - c1[1]);
                const float t3 = c0[2] + 0.25f * (c0[3] - c0[2]), b3 = c1[2] + 0.25f * (c1[3] - c1[2]);
                u[0] = t0 + ty.w1 * (b0 - t0);
                u[1] = t1 + ty.w1 * (b1 - t1);
                u[2] = t2 + ty.w1 * (b2 - t2);
                u[3] = t3 + ty.w1 * (b3 - t3);
            } else {
                const float inv2s = 0.5f / (float)sc;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int nk = nx0 + 2 * k, ik = (nk >> sh1) - bcol;      // 0, 1 or 2
                    const float w1 = (float)(nk & mask) * inv2s;
                    const float a0 = ik == 0 ? c0[0] : (ik == 1 ? c0[1] : c0[2]);
                    const float a1 = ik == 0 ? c0[1] : (ik == 1 ? c0[2] : c0[3]);
                    const float b0 = ik == 0 ? c1[0] : (ik == 1 ? c1[1] : c1[2]);
                    const float b1 = ik == 0 ? c1[1] : (ik == 1 ? c1[2] : c1[3]);
                    const float top = a0 + w1 * (a1 - a0);
                    const float bot = b0 + w1 * (b1 - b0);
                    u[k] = top + ty.w1 * (bot - top);
                }
            }
            t.x += u[0];
            t.y += u[1];
            t.z += u[2];
            t.w += u[3];
        }
    }
    return t;
}

__device__ __forceinline__ float block_sum(float v, float *s_red) {      // 256 threads
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wv] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// part[(n * chunks + chunk)][2C]: (sum t | sum t^2) of the block's pixels, for its channel only; the
// other columns of the row are left untouched, so the partial buffer is zero-filled by the host.
__global__ __launch_bounds__(256) void tail_stats_kernel(Operands o, float *__restrict__ part) {
    __shared__ float s_red[4];
    const int64_t plane = blockIdx.x / o.chunks;
    const int chunk = blockIdx.x % o.chunks;
    const int c = (int)(plane % o.C);
    const int64_t n = plane / o.C;
    const int r0 = chunk * o.rows_per_block, r1 = min(o.H, r0 + o.rows_per_block);
    const int wv4 = o.W >> 2;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < (r1 - r0) * wv4; i += 256) {
        const int iy = wv4 == 1 ? i : (int)__umulhi((unsigned)i, o.quad_magic), y = r0 + iy, x4 = (i - iy * wv4) * 4;
        const float4 t = sum4(o, plane, y, x4);
        s1 += (t.x + t.y) + (t.z + t.w);
        s2 += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    }
    s1 = block_sum(s1, s_red);
    s2 = block_sum(s2, s_red);
    if (threadIdx.x == 0) {
        float *row = part + (n * o.chunks + chunk) * 2 * o.C;
        row[c] = s1;
        row[o.C + c] = s2;
    }
}

__global__ __launch_bounds__(256) void tail_apply_kernel(Operands o, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, void *__restrict__ y) {
    const int64_t plane = blockIdx.x / o.chunks;
    const int chunk = blockIdx.x % o.chunks;
    const int c = (int)(plane % o.C);
    const int r0 = chunk * o.rows_per_block, r1 = min(o.H, r0 + o.rows_per_block);
    const int wv4 = o.W >> 2;
    const float sc = rstd[c] * (gamma ? gamma[c] : 1.f);
    const float sh = (beta ? beta[c] : 0.f) - mean[c] * sc;
    const float lo = o.relu ? 0.f : -INFINITY;
    for (int i = threadIdx.x; i < (r1 - r0) * wv4; i += 256) {
        const int iy = wv4 == 1 ? i : (int)__umulhi((unsigned)i, o.quad_magic), yy = r0 + iy, x4 = (i - iy * wv4) * 4;
        const float4 t = sum4(o, plane, yy, x4);
        store4(y, (plane * o.H + yy) * o.W + x4, o.y_bf16,
               make_float4(fmaxf(t.x * sc + sh, lo), fmaxf(t.y * sc + sh, lo), fmaxf(t.z * sc + sh, lo),
                           fmaxf(t.w * sc + sh, lo)));
    }
}

// Gradient reaching the normalisation output: dy, masked by the fused ReLU (y recomputed, not stored).
__device__ __forceinline__ float4 grad4(const Operands &o, const Raw4 &dy_raw, const float4 &t, float sc, float sh) {
    float4 g = decode(dy_raw, o.dy_bf16);
    if (o.relu) {
        g.x = t.x * sc + sh > 0.f ? g.x : 0.f;
        g.y = t.y * sc + sh > 0.f ? g.y : 0.f;
        g.z = t.z * sc + sh > 0.f ? g.z : 0.f;
        g.w = t.w * sc + sh > 0.f ? g.w : 0.f;
    }
    return g;
}

// part row: (sum dy | sum dy * xhat) for the block's channel.
__global__ __launch_bounds__(256) void tail_bwd_stats_kernel(Operands o, const float *__restrict__ mean,
                                                             const float *__restrict__ rstd,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta,
                                                             const void *__restrict__ dy,
                                                             float *__restrict__ part) {
    __shared__ float s_red[4];
    const int64_t plane = blockIdx.x / o.chunks;
    const int chunk = blockIdx.x % o.chunks;
    const int c = (int)(plane % o.C);
    const int64_t n = plane / o.C;
    const int r0 = chunk * o.rows_per_block, r1 = min(o.H, r0 + o.rows_per_block);
    const int wv4 = o.W >> 2;
    const float mu = mean[c], rs = rstd[c];
    const float sc = rs * (gamma ? gamma[c] : 1.f), sh = (beta ? beta[c] : 0.f) - mu * sc;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < (r1 - r0) * wv4; i += 256) {
        const int iy = wv4 == 1 ? i : (int)__umulhi((unsigned)i, o.quad_magic), yy = r0 + iy, x4 = (i - iy * wv4) * 4;
        Raw4 rdy;
        const float4 t = sum4(o, plane, yy, x4, dy, &rdy);
        const float4 g = grad4(o, rdy, t, sc, sh);
        s1 += (g.x + g.y) + (g.z + g.w);
        s2 += (g.x * (t.x - mu) + g.y * (t.y - mu)) + (g.z * (t.z - mu) + g.w * (t.w - mu));
    }
    s1 = block_sum(s1, s_red);
    s2 = block_sum(s2, s_red) * rs;
    if (threadIdx.x == 0) {
        float *row = part + (n * o.chunks + chunk) * 2 * o.C;
        row[c] = s1;
        row[o.C + c] = s2;
    }
}

// dt = gamma * rstd * (dy - mdy - xhat * mdyx);  da = db = dt (in their own dtypes);
// dx_lo += upsample^T(dt): the tile of dt is staged in LDS and every low-res pixel gathers its
// (2s x 2s) footprint, separably (columns, then rows); rows shared with the neighbouring tile go
// through fp32 atomics on the small low-res map.
__global__ __launch_bounds__(256) void tail_bwd_apply_kernel(Operands o, const float *__restrict__ mean,
                                                             const float *__restrict__ rstd,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta,
                                                             const void *__restrict__ dy,
                                                             const float *__restrict__ mdy,
                                                             const float *__restrict__ mdyx, void *__restrict__ da,
                                                             void *__restrict__ db, float *__restrict__ dxlo) {
    extern __shared__ __attribute__((aligned(16))) float s_tile[];     // [rows][W] dt, then [rows][Wl] column sums
    const int64_t plane = blockIdx.x / o.chunks;
    const int chunk = blockIdx.x % o.chunks;
    const int c = (int)(plane % o.C);
    const int r0 = chunk * o.rows_per_block, r1 = min(o.H, r0 + o.rows_per_block);
    const int rows = r1 - r0;
    const int wv4 = o.W >> 2;
    const float mu = mean[c], rs = rstd[c];
    const float k = rs * (gamma ? gamma[c] : 1.f), m1 = mdy[c], m2 = mdyx[c];
    const float sh = (beta ? beta[c] : 0.f) - mu * k;
    const bool want_lo = dxlo != nullptr && o.x != nullptr;
    for (int i = threadIdx.x; i < rows * wv4; i += 256) {
        const int iy = wv4 == 1 ? i : (int)__umulhi((unsigned)i, o.quad_magic), yy = r0 + iy, x4 = (i - iy * wv4) * 4;
        const int64_t idx = (plane * o.H + yy) * o.W + x4;
        Raw4 rdy;
        const float4 t = sum4(o, plane, yy, x4, dy, &rdy);
        const float4 g = grad4(o, rdy, t, k, sh);
        const float4 d = make_float4(k * (g.x - m1 - (t.x - mu) * rs * m2), k * (g.y - m1 - (t.y - mu) * rs * m2),
                                     k * (g.z - m1 - (t.z - mu) * rs * m2), k * (g.w - m1 - (t.w - mu) * rs * m2));
        if (da) store4(da, idx, o.a_bf16, d);
        if (db) store4(db, idx, o.b_bf16, d);
        if (want_lo) {
            if (o.scale == 1) *reinterpret_cast<float4 *>(dxlo + idx) = d;      // same grid: plain store
            else *reinterpret_cast<float4 *>(s_tile + (yy - r0) * o.W + x4) = d;
        }
    }
    if (!want_lo || o.scale == 1) return;
    __syncthreads();
    const int s = o.scale;
    float *s_col = s_tile + rows * o.W;                                  // [rows][Wl]
    for (int i = threadIdx.x; i < rows * o.Wl; i += 256) {
        const int ry = i / o.Wl, j = i % o.Wl;
        const float *row = s_tile + ry * o.W;
        float acc = 0.f;
        const int xa = max(0, s * (j - 1)), xb = min(o.W - 1, s * (j + 2) - 1);
        for (int x = xa; x <= xb; ++x) {
            const Tap t = tap_of(x, o.Wl, s);
            const float w = (t.i0 == j ? 1.f - t.w1 : 0.f) + (t.i1 == j ? t.w1 : 0.f);
            acc += w * row[x];
        }
        s_col[i] = acc;
    }
    __syncthreads();
    const int i_lo = tap_of(r0, o.Hl, s).i0, i_hi = tap_of(r1 - 1, o.Hl, s).i1;
    float *out = dxlo + plane * o.Hl * o.Wl;
    for (int i = threadIdx.x; i < (i_hi - i_lo + 1) * o.Wl; i += 256) {
        const int li = i_lo + i / o.Wl, j = i % o.Wl;
        const int ya = max(r0, s * (li - 1)), yb = min(r1 - 1, s * (li + 2) - 1);
        float acc = 0.f;
        for (int y = ya; y <= yb; ++y) {
            const Tap t = tap_of(y, o.Hl, s);
            const float w = (t.i0 == li ? 1.f - t.w1 : 0.f) + (t.i1 == li ? t.w1 : 0.f);
            acc += w * s_col[(y - r0) * o.Wl + j];
        }
        // rows whose footprint lies wholly inside this tile are owned by it; the others are shared
        const bool owned = s * (li - 1) >= r0 && s * (li + 2) - 1 <= r1 - 1;
        if (owned || (li == 0 && r0 == 0 && s * (li + 2) - 1 <= r1 - 1) ||
            (li == o.Hl - 1 && r1 == o.H && s * (li - 1) >= r0))
            out[(int64_t)li * o.Wl + j] += acc;                         // no other workgroup touches this row
        else
            atomicAdd(out + (int64_t)li * o.Wl + j, acc);
    }
}

__global__ __launch_bounds__(256) void tail_finalize(const float *__restrict__ part, int nparts, int K,
                                                     float *__restrict__ out) {
    __shared__ float s_acc[8][32];
    const int col = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + col;
    float acc = 0.f;
    if (k < K)
        for (int p = pl; p < nparts; p += 8) acc += part[(int64_t)p * K + k];
    s_acc[pl][col] = acc;
    __syncthreads();
    if (pl == 0 && k < K) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += s_acc[u][col];
        out[k] = t;
    }
}

// sums = [sum t (C) | sum t^2 (C) | count (1)]  ->  mean, rstd (biased variance, as the normalisation
// uses) and the running statistics (unbiased variance, momentum) in ONE small launch; done with
// separate torch ops this bookkeeping was ~10 kernel launches per BatchNorm call.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *__restrict__ sums, int C, float eps, float momentum,
                                                          float *__restrict__ running_mean,
                                                          float *__restrict__ running_var, float *__restrict__ mean,
                                                          float *__restrict__ rstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float count = sums[2 * C];
    const float mu = sums[c] / count;
    const float var = fmaxf(sums[C + c] / count - mu * mu, 0.f);
    mean[c] = mu;
    rstd[c] = rsqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / fmaxf(count - 1.f, 1.f));
    }
}

// (B, T_total, C) fp32 token rows [t0, t0 + T)  <->  (B, C, T) planes (fp32 or bf16), through a
// 32 x 33 LDS tile (both sides coalesced).  TO_PLANES: planes <- tokens; else tokens <- planes
// (+ an optional per-channel vector: conv bias + level embedding of the SPM maps).
template <bool TO_PLANES>
__global__ __launch_bounds__(256) void transpose_tokens_kernel(const void *__restrict__ src, void *__restrict__ dst,
                                                               int64_t T_total, int64_t t0, int T, int C,
                                                               int planes_bf16, const float *__restrict__ vec) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int tt = blockIdx.x * 32, cc = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;          // 32 x 8
    const int64_t tok_off = ((int64_t)b * T_total + t0) * C;
    const int64_t plane_base = (int64_t)b * C * T;
    if (TO_PLANES) {
        const float *tok = reinterpret_cast<const float *>(src) + tok_off;
        float v4[4];                                                  // 4 clamped loads in flight, selected afterwards
#pragma unroll
        for (int u = 0; u < 4; ++u)                                   // tile[token][channel], channel fastest
            v4[u] = tok[(int64_t)min(tt + ly + 8 * u, T - 1) * C + min(cc + lx, C - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) tile[ly + 8 * u][lx] = (tt + ly + 8 * u < T && cc + lx < C) ? v4[u] : 0.f;
        __syncthreads();
#pragma unroll
        for (int r = ly; r < 32; r += 8)                              // write plane rows: token fastest
            if (cc + r < C && tt + lx < T) {
                const int64_t o = plane_base + (int64_t)(cc + r) * T + tt + lx;
                if (planes_bf16) reinterpret_cast<__bf16 *>(dst)[o] = (__bf16)tile[lx][r];
                else reinterpret_cast<float *>(dst)[o] = tile[lx][r];
            }
    } else {
        float v4[4];
        int64_t o4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)                                   // tile[channel][token], token fastest
            o4[u] = plane_base + (int64_t)min(cc + ly + 8 * u, C - 1) * T + min(tt + lx, T - 1);
        if (planes_bf16) {                                            // dtype branch around the 4 loads, not inside
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = (float)reinterpret_cast<const __bf16 *>(src)[o4[u]];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = reinterpret_cast<const float *>(src)[o4[u]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) tile[ly + 8 * u][lx] = (cc + ly + 8 * u < C && tt + lx < T) ? v4[u] : 0.f;
        __syncthreads();
        float *tok = reinterpret_cast<float *>(dst) + tok_off;
        const float add = (vec && cc + lx < C) ? vec[cc + lx] : 0.f;
#pragma unroll
        for (int r = ly; r < 32; r += 8)
            if (tt + r < T && cc + lx < C) tok[(int64_t)(tt + r) * C + cc + lx] = tile[lx][r] + add;
    }
}

// MaxPool2d(kernel 3, stride 2, padding 1) of the SPM stem (adapter_modules.py:229-230), bf16 NCHW.
// Forward keeps the window position (0..8) of the FIRST maximum in row-major scan order - the element
// torch's max_pool2d routes the gradient to - in one byte per output; backward is a gather: every
// input pixel lies in at most 4 windows and takes their gradients where it is the recorded maximum
// (torch's backward scatters with atomics: 222 us for the 2 x 64 x 512 x 512 stem map).
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const __bf16 *__restrict__ x, int H, int W, int Ho, int Wo,
                                                             __bf16 *__restrict__ y, unsigned char *__restrict__ idx) {
    // block = 64 output columns x 4 output rows of one plane (blockIdx.z): no index divisions
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= Wo || oy >= Ho) return;
    const int64_t plane = blockIdx.z;
    const __bf16 *xp = x + plane * H * W;
    // the 9 window values are requested together (clamped coordinates), validity applied afterwards: loads
    // under `if (inside)` were waited on one by one
    float win[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int iy = min(max(2 * oy - 1 + k / 3, 0), H - 1), ix = min(max(2 * ox - 1 + k % 3, 0), W - 1);
        win[k] = (float)xp[(int64_t)iy * W + ix];
    }
    __builtin_amdgcn_sched_barrier(0);
    float best = -INFINITY;
    int pos = 0;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float v = win[k];
        if (in && (!any || v > best || v != v)) {
            best = v;
            pos = k;
            any = true;
        }
    }
    const int64_t o = (plane * Ho + oy) * Wo + ox;
    y[o] = (__bf16)best;
    idx[o] = (unsigned char)pos;
}

// thread = 8 consecutive input columns of one input row (one 16-byte store); block = 64 x 4 such threads
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const __bf16 *__restrict__ gy,
                                                             const unsigned char *__restrict__ idx, int H, int W, int Ho,
                                                             int Wo, __bf16 *__restrict__ gx) {
    const int ix0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ix0 >= W || iy >= H) return;
    const int64_t plane = blockIdx.z;
    const __bf16 *gp = gy + plane * Ho * Wo;
    const unsigned char *ip = idx + plane * Ho * Wo;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1;                  // the one or two window rows that contain iy
    const int oxb = ix0 >> 1;                                       // windows oxb .. oxb + 4 touch these 8 columns
    // all 20 loads (2 window rows x 5 windows x {gradient, recorded position}) are requested first, with
    // clamped coordinates; validity is applied afterwards (loads under `in ? .. : ..` were waited on in pairs)
    float g[2][5];
    int p[2][5];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oyc = min(a == 0 ? oy0 : oy1, Ho - 1);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int64_t at = (int64_t)oyc * Wo + min(oxb + u, Wo - 1);
            g[a][u] = (float)gp[at];
            p[a][u] = (int)ip[at];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = a == 0 ? oy0 : oy1;
        if ((a == 1 && oy1 == oy0) || oy >= Ho) continue;
        const int ky3 = (iy - (2 * oy - 1)) * 3;
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (oxb + u >= Wo) p[a][u] = -1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // column ix0 + e lies in window (ix0 + e) >> 1 at kx = 1 (even e) or 2 (odd e), and in the next
            // window at kx = 0 (odd e only); ix0 is a multiple of 8
            const int u0 = e >> 1;
            const int kx0 = (e & 1) ? 2 : 1;
            if (p[a][u0] == ky3 + kx0) acc[e] += g[a][u0];
            if (e & 1)
                if (p[a][u0 + 1] == ky3 + 0) acc[e] += g[a][u0 + 1];
        }
    }
    __bf16 *o = gx + (plane * H + iy) * (int64_t)W + ix0;
    if (ix0 + 8 <= W && (W & 7) == 0) {
        typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
        bf16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)acc[e];
        *reinterpret_cast<bf16x8_t *>(o) = v;
    } else {
        for (int e = 0; e < 8 && ix0 + e < W; ++e) o[e] = (__bf16)acc[e];
    }
}

int fill_operands(const char *fn, Operands &o, const void *a, int a_bf16, const void *b, int b_bf16, const float *x,
                  int scale, int64_t N, int64_t C, int64_t H, int64_t W) {
    if (N < 1 || C < 1 || H < 1 || W < 4 || W % 4 || W > kTilePx) return fail(VAH_E_SHAPE, "%s: bad shape", fn);
    if (scale != 1 && scale != 2 && scale != 4 && scale != 8) return fail(VAH_E_SHAPE, "%s: scale must be 1, 2, 4 or 8", fn);
    if (x && (H % scale || W % scale || (W / scale) % 4)) return fail(VAH_E_SHAPE, "%s: H, W not multiples of the scale", fn);
    if (N * C * H * W >= ((int64_t)1 << 40)) return fail(VAH_E_SHAPE, "%s: too large", fn);
    if (!a) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)a | (uintptr_t)b) % 8 || (uintptr_t)x % 16 || (!a_bf16 && (uintptr_t)a % 16) || (b && !b_bf16 && (uintptr_t)b % 16))
        return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    o.relu = o.y_bf16 = o.dy_bf16 = 0;
    o.shift = nullptr;
    o.a = a;
    o.b = b;
    o.x = x;
    o.a_bf16 = a_bf16;
    o.b_bf16 = b_bf16;
    o.scale = scale;
    o.C = (int)C;
    o.H = (int)H;
    o.W = (int)W;
    o.Hl = (int)(H / scale);
    o.Wl = (int)(W / scale);
    int rpb = std::max<int>(1, kTilePx / (int)W);
    rpb = std::max(scale * 2, rpb / (scale * 2) * (scale * 2));        // multiple of 2s: tiles end between footprints
    rpb = std::min<int>(rpb, (int)H);
    o.rows_per_block = rpb;
    o.chunks = (int)((H + rpb - 1) / rpb);
    if (N * o.chunks > kTailParts) {                                     // keep the partial rows bounded
        o.chunks = std::max<int>(1, kTailParts / (int)N);
        o.rows_per_block = (int)((H + o.chunks - 1) / o.chunks);
        o.rows_per_block = (o.rows_per_block + 2 * scale - 1) / (2 * scale) * (2 * scale);
        o.chunks = (int)((H + o.rows_per_block - 1) / o.rows_per_block);
    }
    // exact for items < 2^16 (a block has rows_per_block * W / 4 <= 8192 / 4 + 2 * scale * W / 4 of them)
    if ((int64_t)o.rows_per_block * (W / 4) >= 65536) return fail(VAH_E_SHAPE, "%s: row too wide for the tiling", fn);
    o.quad_magic = W / 4 > 1 ? (unsigned)(((uint64_t)1 << 32) / (uint64_t)(W / 4)) + 1u : 0u;      // one quad per row: no division
    return VAH_OK;
}


// ConvTranspose2d(k = 2, stride 2) as GEMMs on token rows: the product U (B, 4 * C, h * w), rows (dy, dx, co), is the
// transposed convolution's output with its 2 x 2 sub-pixels still apart; this pass interleaves them into NCHW planes
//   out[b][co][2 y + dy][2 x + dx] = U[b][(2 dy + dx) * C + co][y * w + x]            (inverse: the other way round)
// One thread: 8 consecutive x of one source row pair (dx = 0, 1) <-> 16 consecutive output pixels (16-byte accesses).
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 ps_bf16x8;
// `add` (forward only, planes-shaped bf16 or null): dst = interleave(src) + add, rounded once - the `up(c2) + c1` of
// vit_adapter.py:107 in the precision autocast gives it (two bf16 tensors summed into bf16)
template <bool INVERSE>
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const __bf16 *__restrict__ src, __bf16 *__restrict__ dst,
                                                             const __bf16 *__restrict__ add, int B, int C, int h, int w,
                                                             int64_t items) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (b, co, Y, x8)
    if (i >= items) return;
    const int w8 = w >> 3;
    const int x8 = (int)(i % w8);
    int64_t t = i / w8;
    const int Y = (int)(t % (2 * h));
    t /= 2 * h;
    const int co = (int)(t % C), b = (int)(t / C);
    const int y = Y >> 1, dy = Y & 1;
    const int64_t u0 = (((int64_t)b * 4 + 2 * dy) * C + co) * ((int64_t)h * w) + (int64_t)y * w + 8 * x8;       // dx = 0 row
    const int64_t u1 = u0 + (int64_t)C * h * w;                                                                  // dx = 1 row
    const int64_t o = (((int64_t)b * C + co) * (2 * h) + Y) * (2 * (int64_t)w) + 16 * x8;
    if (!INVERSE) {
        const ps_bf16x8 a = *reinterpret_cast<const ps_bf16x8 *>(src + u0), c = *reinterpret_cast<const ps_bf16x8 *>(src + u1);
        ps_bf16x8 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[2 * j] = a[j], lo[2 * j + 1] = c[j];
            hi[2 * j] = a[4 + j], hi[2 * j + 1] = c[4 + j];
        }
        if (add) {
            const ps_bf16x8 p = *reinterpret_cast<const ps_bf16x8 *>(add + o), q = *reinterpret_cast<const ps_bf16x8 *>(add + o + 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lo[j] = (__bf16)((float)lo[j] + (float)p[j]);
                hi[j] = (__bf16)((float)hi[j] + (float)q[j]);
            }
        }
        *reinterpret_cast<ps_bf16x8 *>(dst + o) = lo;
        *reinterpret_cast<ps_bf16x8 *>(dst + o + 8) = hi;
    } else {
        const ps_bf16x8 lo = *reinterpret_cast<const ps_bf16x8 *>(src + o), hi = *reinterpret_cast<const ps_bf16x8 *>(src + o + 8);
        ps_bf16x8 a, c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = lo[2 * j], c[j] = lo[2 * j + 1];
            a[4 + j] = hi[2 * j], c[4 + j] = hi[2 * j + 1];
        }
        *reinterpret_cast<ps_bf16x8 *>(dst + u0) = a;
        *reinterpret_cast<ps_bf16x8 *>(dst + u1) = c;
    }
}

}  // namespace
}  // namespace vah

extern "C" {

int64_t vah_bn_tail_ws_floats(int64_t C) { return (int64_t)vah::kTailParts * 2 * C; }

int vah_bn_tail_stats(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale, int64_t N,
                      int64_t C, int64_t H, int64_t W, const float *shift, float *sums, float *ws, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_tail_stats";
    Operands o;
    if (int rc = fill_operands(fn, o, a, a_bf16, b, b_bf16, x, scale, N, C, H, W)) return rc;
    if (!sums || !ws) return fail(VAH_E_NULL, "%s: null pointer", fn);
    o.shift = shift;
    hipStream_t st = (hipStream_t)stream;
    const int nparts = (int)N * o.chunks;
    LaunchScope scope("bn_tail_stats", N * C * H * W * ((a_bf16 ? 2 : 4) + (b ? (b_bf16 ? 2 : 4) : 0)), st);
    hipLaunchKernelGGL(tail_stats_kernel, dim3((unsigned)(N * C * o.chunks)), dim3(256), 0, st, o, ws);
    hipLaunchKernelGGL(tail_finalize, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), 0, st, ws, nparts, (int)(2 * C), sums);
    return check_launch(fn);
}

int vah_bn_tail_apply(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale, int64_t N,
                      int64_t C, int64_t H, int64_t W, const float *mean, const float *rstd, const float *gamma,
                      const float *beta, int relu, const float *shift, void *y, int y_bf16, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_tail_apply";
    Operands o;
    if (int rc = fill_operands(fn, o, a, a_bf16, b, b_bf16, x, scale, N, C, H, W)) return rc;
    if (!mean || !rstd || !y) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if ((uintptr_t)y % (y_bf16 ? 8 : 16)) return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    o.relu = relu != 0;
    o.y_bf16 = y_bf16 != 0;
    o.shift = shift;
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("bn_tail_apply", N * C * H * W * ((a_bf16 ? 2 : 4) + (b ? (b_bf16 ? 2 : 4) : 0) + 4), st);
    hipLaunchKernelGGL(tail_apply_kernel, dim3((unsigned)(N * C * o.chunks)), dim3(256), 0, st, o, mean, rstd, gamma, beta, y);
    return check_launch(fn);
}

int vah_bn_tail_bwd_stats(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale, int64_t N,
                          int64_t C, int64_t H, int64_t W, const float *mean, const float *rstd, const float *gamma,
                          const float *beta, int relu, const float *shift, const void *dy, int dy_bf16, float *sums,
                          float *ws, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_tail_bwd_stats";
    Operands o;
    if (int rc = fill_operands(fn, o, a, a_bf16, b, b_bf16, x, scale, N, C, H, W)) return rc;
    if (!mean || !rstd || !dy || !sums || !ws) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if ((uintptr_t)dy % (dy_bf16 ? 8 : 16)) return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    o.relu = relu != 0;
    o.dy_bf16 = dy_bf16 != 0;
    o.shift = shift;
    hipStream_t st = (hipStream_t)stream;
    const int nparts = (int)N * o.chunks;
    LaunchScope scope("bn_tail_bwd_stats", N * C * H * W * ((a_bf16 ? 2 : 4) + (b ? (b_bf16 ? 2 : 4) : 0) + 4), st);
    hipLaunchKernelGGL(tail_bwd_stats_kernel, dim3((unsigned)(N * C * o.chunks)), dim3(256), 0, st, o, mean, rstd, gamma, beta,
                       dy, ws);
    hipLaunchKernelGGL(tail_finalize, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), 0, st, ws, nparts, (int)(2 * C), sums);
    return check_launch(fn);
}

int vah_bn_tail_bwd_apply(const void *a, int a_bf16, const void *b, int b_bf16, const float *x, int scale, int64_t N,
                          int64_t C, int64_t H, int64_t W, const float *mean, const float *rstd, const float *gamma,
                          const float *beta, int relu, const float *shift, const void *dy, int dy_bf16, const float *mdy,
                          const float *mdyx, void *da, void *db, float *dxlo, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_tail_bwd_apply";
    Operands o;
    if (int rc = fill_operands(fn, o, a, a_bf16, b, b_bf16, x, scale, N, C, H, W)) return rc;
    if (!mean || !rstd || !dy || !mdy || !mdyx) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if ((uintptr_t)dy % (dy_bf16 ? 8 : 16) || (uintptr_t)dxlo % 16 || ((uintptr_t)da | (uintptr_t)db) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    o.relu = relu != 0;
    o.dy_bf16 = dy_bf16 != 0;
    o.shift = shift;
    hipStream_t st = (hipStream_t)stream;
    size_t smem = 0;
    if (dxlo && x && scale > 1) smem = (size_t)o.rows_per_block * (o.W + o.Wl) * sizeof(float);
    if (smem > 150 * 1024) return fail(VAH_E_SHAPE, "%s: tile does not fit LDS", fn);
    LaunchScope scope("bn_tail_bwd_apply", N * C * H * W * (2 * ((a_bf16 ? 2 : 4) + (b ? (b_bf16 ? 2 : 4) : 0)) + 4), st);
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)tail_bwd_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(tail_bwd_apply_kernel, dim3((unsigned)(N * C * o.chunks)), dim3(256), smem, st, o, mean, rstd, gamma,
                       beta, dy, mdy, mdyx, da, db, dxlo);
    return check_launch(fn);
}

// sums = [sum (C) | sum of squares (C) | element count (1)] (as written by vah_bn_tail_stats plus the
// count, all-reduced over the ranks for SyncBatchNorm) -> mean, rstd; running statistics updated in
// place when given (momentum, unbiased variance), as torch.nn.BatchNorm does in training.
int vah_bn_finalize_stats(const float *sums, int64_t C, float eps, float momentum, float *running_mean,
                          float *running_var, float *mean, float *rstd, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_finalize_stats";
    if (C < 1) return fail(VAH_E_SHAPE, "%s: bad C", fn);
    if (!sums || !mean || !rstd || ((running_mean != nullptr) != (running_var != nullptr))) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sums, (int)C, eps,
                       momentum, running_mean, running_var, mean, rstd);
    return check_launch(fn);
}

// tokens -> planes (to_planes != 0): dst (B, C, T) <- src (B, T_total, C)[:, t0 : t0 + T, :];
// planes -> tokens: dst (B, T_total, C)[:, t0 : t0 + T, :] <- src (B, C, T) (+ vec[C] if given).
// Tokens are fp32; the planes fp32 or bf16 (planes_bf16).
// Replaces  c[:, a:b].transpose(1, 2).view(B, C, H, W).contiguous()  of the pyramid assembly
// (vit_adapter.py:113-119) and  cat([fc_l(c_l).flatten(2).transpose(1, 2) + level_embed[l]])  of
// vit_adapter.py:94-97 / adapter_modules.py:262-268, and their backward passes.
int vah_transpose_tokens(const void *src, int64_t B, int64_t T_total, int64_t t0, int64_t T, int64_t C, void *dst,
                         int to_planes, int planes_bf16, const float *vec, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_transpose_tokens";
    if (B < 0 || T_total < 0 || t0 < 0 || T < 0 || t0 + T > T_total || C < 1 || B > 65535 || C > 65535 * 32 || T >= ((int64_t)1 << 31))
        return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (B == 0 || T == 0) return VAH_OK;
    if (!src || !dst) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (to_planes && vec) return fail(VAH_E_UNSUPPORTED, "%s: vec only applies to planes -> tokens", fn);
    const dim3 grid((unsigned)((T + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("transpose_tokens", B * T * C * (4 + (planes_bf16 ? 2 : 4)), st);
    if (to_planes)
        hipLaunchKernelGGL(transpose_tokens_kernel<true>, grid, dim3(256), 0, st, src, dst, T_total, t0, (int)T, (int)C,
                           planes_bf16, vec);
    else
        hipLaunchKernelGGL(transpose_tokens_kernel<false>, grid, dim3(256), 0, st, src, dst, T_total, t0, (int)T, (int)C,
                           planes_bf16, vec);
    return check_launch(fn);
}

// MaxPool2d(3, stride 2, padding 1) on bf16 NCHW (planes = N * C): y, idx (1 byte per output: window position
// of the first maximum) <- x;  backward: gx <- gy, idx (gather, no atomics).
int vah_maxpool3s2_fwd_bf16(const void *x, int64_t planes, int64_t H, int64_t W, void *y, void *idx, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_maxpool3s2_fwd_bf16";
    if (planes < 0 || H < 1 || W < 1 || H > (1 << 20) || W > (1 << 20)) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    const int64_t Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, total = planes * Ho * Wo;
    if (total == 0) return VAH_OK;
    if (!x || !y || !idx) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("maxpool_fwd", planes * (H * W * 2 + Ho * Wo * 3), st);
    if (planes > 65535 || (Ho + 3) / 4 > 65535) return fail(VAH_E_SHAPE, "%s: too many planes / rows", fn);
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3((unsigned)((Wo + 63) / 64), (unsigned)((Ho + 3) / 4), (unsigned)planes),
                       dim3(256), 0, st, (const __bf16 *)x, (int)H, (int)W, (int)Ho, (int)Wo, (__bf16 *)y,
                       (unsigned char *)idx);
    return check_launch(fn);
}

int vah_maxpool3s2_bwd_bf16(const void *gy, const void *idx, int64_t planes, int64_t H, int64_t W, void *gx, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_maxpool3s2_bwd_bf16";
    if (planes < 0 || H < 1 || W < 1 || H > (1 << 20) || W > (1 << 20)) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    const int64_t Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, total = planes * H * W;
    if (total == 0) return VAH_OK;
    if (!gy || !idx || !gx) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("maxpool_bwd", planes * (H * W * 2 + Ho * Wo * 3), st);
    if (planes > 65535 || (H + 3) / 4 > 65535) return fail(VAH_E_SHAPE, "%s: too many planes / rows", fn);
    if ((uintptr_t)gx % 16) return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((unsigned)((W + 511) / 512), (unsigned)((H + 3) / 4), (unsigned)planes),
                       dim3(256), 0, st, (const __bf16 *)gy, (const unsigned char *)idx, (int)H, (int)W, (int)Ho, (int)Wo,
                       (__bf16 *)gx);
    return check_launch(fn);
}


/* inverse == 0: planes (B, C, 2h, 2w) <- U (B, 4*C, h*w), rows (dy, dx, co) [+ add, planes-shaped, optional];
 * inverse != 0: U <- planes.  bf16, w % 8 == 0. */
int vah_pixel_shuffle2_bf16(const void *src, int64_t B, int64_t C, int64_t h, int64_t w, void *dst, int inverse, const void *add,
                            void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_pixel_shuffle2_bf16";
    if (B < 0 || C < 1 || h < 1 || w < 8 || w % 8) return fail(VAH_E_SHAPE, "%s: w must be a multiple of 8", fn);
    if (B == 0) return VAH_OK;
    if (!src || !dst) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)src | (uintptr_t)dst) % 16) return fail(VAH_E_ALIGN, "%s: 16-byte alignment", fn);
    const int64_t items = B * C * 2 * h * (w / 8);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("pixel_shuffle2", B * C * 4 * h * w * 4, st);
    if (add && (inverse || (uintptr_t)add % 16)) return fail(VAH_E_SHAPE, "%s: add is a forward-only, 16-byte aligned operand", fn);
    if (inverse)
        hipLaunchKernelGGL(pixel_shuffle2_kernel<true>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (const __bf16 *)src,
                           (__bf16 *)dst, (const __bf16 *)nullptr, (int)B, (int)C, (int)h, (int)w, items);
    else
        hipLaunchKernelGGL(pixel_shuffle2_kernel<false>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (const __bf16 *)src,
                           (__bf16 *)dst, (const __bf16 *)add, (int)B, (int)C, (int)h, (int)w, items);
    return check_launch(fn);
}

}  // extern "C"
