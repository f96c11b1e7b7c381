// Atomic-free grad_value of multi-scale deformable attention for gfx950: on-device binning of the
// samples by destination tile + one workgroup per (batch, head, 8x8-pixel tile of one level) that
// sums the tile's contributions on the matrix cores and STORES the tile (no zero-fill, no atomics).
//
// Replaces the scatter of /root/reference/detection/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159
// (ms_deform_attn_col2im_bilinear: 4 atomicAdd per sample and channel, called from :301-403) for ANY
// sampling locations - the plain MSDeformAttnFunction (no reference grid, e.g. the Mask2Former pixel
// decoder with per-batch reference points, seg/.../msdeformattn_pixel_decoder.py:224-242) as well as
// the fused MSDeformAttn core.  d(loc) / d(attn) stay in the gather kernels of msda.hip /
// msda_fused.hip (run with their scatter switched off).
//
// Why this shape (measured on MI355X, tools/ubench/lds_atomic.hip, profiles/r02_lds_atomic_ubench.txt):
//   * memory-side float atomics top out at ~1.3 TB/s of added bytes: the plain backward moved 604 MB
//     (injector) / 1057 MB (extractor) of them: 464 / 837 us;
//   * LDS float atomics are worse: ds_add_f32 retires ONE LANE PER ~3 CLOCKS per CU (0.33 lane-ops/clk,
//     whatever the occupancy or the bank pattern), ds_pk_add_bf16 twice that - 40x below ds_write_b32;
//   * so the tile sum is a dense product instead: per chunk of 64 list entries (one per lane),
//       dV[64 px, 32 ch] += Wt[64 px, 64 entries] x G[64 entries, 32 ch]
//     where lane k builds column k of Wt (its entry's <= 16 attention x bilinear weights that land in
//     the tile) with plain LDS read-add-write on its OWN column, and G holds the entries' grad_out rows.
//     bf16 grad_out: v_mfma_f32_32x32x16_bf16 with Wt split into bf16 hi + lo (~16 mantissa bits) and the
//     G fragment read with ds_read_b64_tr_b16; fp32 grad_out: v_mfma_f32_32x32x2_f32 (exact fp32 fma
//     chains, 1/16 of the bf16 rate and still ~20 us for the largest call).
//
// Passes (all on the caller's stream, workspace from the caller, no host sync):
//   1. bin  : one thread per (n, m, level, q): the tiles its P = 4 samples touch; q is appended to their lists
//             (per-workgroup counting in LDS, one returning global atomic per list and workgroup).  Lists have
//             a fixed capacity per level (a multiple of the load of evenly spread samples); a list that
//             overflows is simply not used:
//   2. tile : one single-wave workgroup per list (13 KB of LDS: 12 per CU), lists of the 12 heads of one tile
//             adjacent (they read the same query rows: L2 hits instead of 64-byte pieces of 128-byte lines
//             from HBM); the wave walks its list in chunks of 64 entries.  A workgroup whose list overflowed walks ALL queries of its
//             (n, head, level) instead and masks them to its tile - slow (Lq / 64 chunks) but exact, and
//             only reached by sampling patterns that pile > 4x the mean load onto one tile.
// A list entry is 4 bytes (q); the tile kernel re-reads the entry's locations / weights / grad_out row.
#include <cstdlib>
#include <type_traits>

#include "msda_common.h"
#include "msda_internal.h"

namespace vah {
namespace {

using namespace vah::msda;

constexpr int kTW = 8, kTH = 4;    // tile: 8 x 4 pixels = the 32 rows of one 32x32 MFMA tile
constexpr int kTShY = 2, kTShX = 3;
constexpr int kTilePx = kTW * kTH;
constexpr int kP = 4;              // points per level on this path
constexpr int kMaxL = 4;
constexpr int kD = 32;
constexpr int kTileThreads = 64;   // ONE wave per tile workgroup: 13 KB of LDS, 12 workgroups per CU

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;

struct TileGeom {                  // host copy of the level geometry (kernel argument: scalar loads)
    int L, T;                      // levels, tiles per (n, m)
    int H[kMaxL], W[kMaxL], start[kMaxL], ntx[kMaxL], tbase[kMaxL];
    int cap[kMaxL], ebase[kMaxL];  // list capacity of the level; first entry slot of the level's lists (per n and head)
    int ET;                        // entry slots per (n, head)
    // tile pass: a list of level l is shared by ksplit[l] single-wave workgroups (1 for almost every level of the
    // adapter's calls; a coarse level - few tiles, every query in each - gets more); the levels are launched longest
    // lists first: wgorder[i] = level of the i-th block of workgroups, wgbase[i] = its first workgroup;
    // slab[l] = first partial-sum slab of level l (in 32 x 32 float tiles; levels with ksplit 1: none)
    int ksplit[kMaxL], ntiles[kMaxL], wgorder[kMaxL];
    long long wgbase[kMaxL + 1], slab[kMaxL];
};

// list (n, tile t of level l, head m): index ((n * T + t) * M + m); its entries start at
// ((n * ET + ebase[l] + (t - tbase[l]) * cap[l]) * M + m * cap[l]
__device__ __forceinline__ int64_t list_entry_base(const TileGeom &g, int64_t n, int l, int t_all, int M, int m) {
    return ((int64_t)n * g.ET + g.ebase[l] + (int64_t)(t_all - g.tbase[l]) * g.cap[l]) * M + (int64_t)m * g.cap[l];
}

// ---- sample sources ---------------------------------------------------------------------------
// A source hands out one (n, q, m) row's samples of one level in two steps: load() only issues the
// global loads - whole 16 / 8-byte vectors, kept as raw words so that a caller can hold the NEXT chunk's
// operands in few registers while it works on the current one - and xy() / weights() decode them.
// Plain: sampling_locations (N,Lq,M,L,P,2) and attention_weights (N,Lq,M,L,P), fp32 (the reference API).
struct PlainSrc {
    const float *loc, *attn;
    int LP;
    float *grad_loc, *grad_attn;           // outputs of the tile pass (spec cuh:156-158)
    // gradients of the P samples of (row, level): own = bit p set when this caller owns sample p
    __device__ __forceinline__ void store_grads(int64_t row, int l, unsigned own, const float (&gx)[kP], const float (&gy)[kP],
                                                const float (&ga)[kP], int H, int W) const {
        float *gl = grad_loc + (row * LP + l * kP) * 2;
        float *gt = grad_attn + row * LP + l * kP;
        if (own == 0xFu) {
            *reinterpret_cast<float4 *>(gl) = make_float4((float)W * gx[0], (float)H * gy[0], (float)W * gx[1], (float)H * gy[1]);
            *reinterpret_cast<float4 *>(gl + 4) = make_float4((float)W * gx[2], (float)H * gy[2], (float)W * gx[3], (float)H * gy[3]);
            *reinterpret_cast<float4 *>(gt) = make_float4(ga[0], ga[1], ga[2], ga[3]);
        } else {
#pragma unroll
            for (int p = 0; p < kP; ++p)
                if ((own >> p) & 1) {
                    *reinterpret_cast<float2 *>(gl + 2 * p) = make_float2((float)W * gx[p], (float)H * gy[p]);
                    gt[p] = ga[p];
                }
        }
    }
    struct Raw {
        float4 xy[2];       // P = 4 locations
        float4 a;
    };
    template <bool WEIGHTS>
    __device__ __forceinline__ Raw load(int64_t row, int64_t q, int l) const {
        Raw r;
        const float4 *lp = reinterpret_cast<const float4 *>(loc + (row * LP + l * kP) * 2);       // 32-byte aligned
        r.xy[0] = lp[0];
        r.xy[1] = lp[1];
        r.a = WEIGHTS ? *reinterpret_cast<const float4 *>(attn + row * LP + l * kP) : make_float4(0.f, 0.f, 0.f, 0.f);
        return r;
    }
    __device__ __forceinline__ float2 xy(const Raw &r, int p, int H, int W) const {
        const float4 v = r.xy[p >> 1];
        return (p & 1) ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
    }
    __device__ __forceinline__ void weights(const Raw &r, int l, float (&a)[kP]) const {
        a[0] = r.a.x, a[1] = r.a.y, a[2] = r.a.z, a[3] = r.a.w;
    }
};

template <typename PT>
__device__ __forceinline__ float word_elem(const uint32_t *w, int i) {       // element i of a packed PT array
    if constexpr (sizeof(PT) == 4) return __builtin_bit_cast(float, w[i]);
    else return __builtin_bit_cast(float, (i & 1) ? (w[i >> 1] & 0xFFFF0000u) : (w[i >> 1] << 16));     // bf16 -> f32
}

// Fused: raw sampling_offsets / attention logits of the MSDeformAttn module + the reference grid
// (arithmetic of msda_fused.hip: loc = ref + off / (W, H), softmax over the L*P logits).
template <typename PT, int L>
struct FusedSrc {
    const PT *off, *logit;
    const float *ref;
    int ref_levels;
    static constexpr int LP = L * kP;
    static constexpr int OW = kP * 2 * (int)sizeof(PT) / 4;      // words of one level's offsets (4 or 8)
    static constexpr int LW = LP * (int)sizeof(PT) / 4;          // words of the row's logits (even)
    PT *d_off;                             // output: d(offsets) = attention x d(out)/d(pixel position) (the level size cancels)
    float *ga;                             // scratch (N,Lq,M,L*P) fp32: d(out)/d(attention probability), for the softmax backward
    __device__ __forceinline__ void store_grads(int64_t row, int l, unsigned own, const float (&gx)[kP], const float (&gy)[kP],
                                                const float (&gav)[kP], int H, int W) const {
        PT *dp = d_off + (row * LP + l * kP) * 2;
        float *gp = ga + row * LP + l * kP;
        if (own == 0xFu) {
            if constexpr (sizeof(PT) == 2) {
                bf16x8 o;
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    o[2 * p] = (__bf16)gx[p];
                    o[2 * p + 1] = (__bf16)gy[p];
                }
                *reinterpret_cast<bf16x8 *>(dp) = o;
            } else {
                *reinterpret_cast<float4 *>(dp) = make_float4(gx[0], gy[0], gx[1], gy[1]);
                *reinterpret_cast<float4 *>(dp + 4) = make_float4(gx[2], gy[2], gx[3], gy[3]);
            }
            *reinterpret_cast<float4 *>(gp) = make_float4(gav[0], gav[1], gav[2], gav[3]);
        } else {
#pragma unroll
            for (int p = 0; p < kP; ++p)
                if ((own >> p) & 1) {
                    if constexpr (sizeof(PT) == 2) {
                        bf16x2 o;
                        o[0] = (__bf16)gx[p];
                        o[1] = (__bf16)gy[p];
                        *reinterpret_cast<bf16x2 *>(dp + 2 * p) = o;
                    } else {
                        *reinterpret_cast<float2 *>(dp + 2 * p) = make_float2(gx[p], gy[p]);
                    }
                    gp[p] = gav[p];
                }
        }
    }
    struct Raw {
        uint32_t o[OW];
        uint32_t lg[LW];
        float2 rp;
    };
    template <bool WEIGHTS>
    __device__ __forceinline__ Raw load(int64_t row, int64_t q, int l) const {
        Raw r;
        const uint4 *op = reinterpret_cast<const uint4 *>(off + (row * LP + l * kP) * 2);         // 16-byte aligned
#pragma unroll
        for (int i = 0; i < OW / 4; ++i) {
            const uint4 v = op[i];
            r.o[4 * i] = v.x, r.o[4 * i + 1] = v.y, r.o[4 * i + 2] = v.z, r.o[4 * i + 3] = v.w;
        }
        r.rp = *reinterpret_cast<const float2 *>(ref + (q * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
        const uint2 *lp = reinterpret_cast<const uint2 *>(logit + row * LP);                       // 8-byte aligned
#pragma unroll
        for (int i = 0; i < LW / 2; ++i) {
            const uint2 v = WEIGHTS ? lp[i] : make_uint2(0, 0);
            r.lg[2 * i] = v.x, r.lg[2 * i + 1] = v.y;
        }
        return r;
    }
    __device__ __forceinline__ float2 xy(const Raw &r, int p, int H, int W) const {
        const float ox = word_elem<PT>(r.o, 2 * p), oy = word_elem<PT>(r.o, 2 * p + 1);
        return make_float2(r.rp.x + ox / (float)W, r.rp.y + oy / (float)H);
    }
    __device__ __forceinline__ void weights(const Raw &r, int l, float (&a)[kP]) const {
        float pr[LP];
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < LP; ++s) {
            pr[s] = word_elem<PT>(r.lg, s);
            mx = fmaxf(mx, pr[s]);
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < LP; ++s) {
            pr[s] = __expf(pr[s] - mx);
            sum += pr[s];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            float v = 0.f;
#pragma unroll
            for (int s = 0; s < LP; ++s) v = (s == l * kP + p) ? pr[s] : v;      // l is not a compile-time constant
            a[p] = v * inv;
        }
    }
};

// One sample's base pixel, fractions and gate: the arithmetic of msda_common.h::make_tap
// (spec cuh:253-266, 288): corner (dy, dx) is pixel (y0 + dy, x0 + dx), valid when `inside` and in the map.
struct Base {
    int y0, x0;
    float lh, lw;
    bool inside;
};
__device__ __forceinline__ Base make_base(float lx, float ly, int H, int W) {
    Base b;
    const float h_im = ly * (float)H - 0.5f;
    const float w_im = lx * (float)W - 0.5f;
    b.inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hs = b.inside ? h_im : 0.f;
    const float ws = b.inside ? w_im : 0.f;
    const float hf = floorf(hs), wf = floorf(ws);
    b.y0 = (int)hf;
    b.x0 = (int)wf;
    b.lh = hs - hf;
    b.lw = ws - wf;
    return b;
}

// ---- pass 1: binning ----------------------------------------------------------------------------
// grid (ceil(Lq / 256), N * M * L): a workgroup is 256 neighbouring queries of one (n, m, level).
// Per thread: the tiles its 4 samples touch - per sample the 1 x 1 .. 2 x 2 block of tiles its corners fall in,
// minus what an earlier sample of the row already named (<= 16 tiles, typically 1 or 2: the samples of a row sit
// within a few pixels of each other in the adapter; spread over the whole map - the `uniform` recipe of the
// reference's test.py - it is 4 to 16).  The workgroup counts per tile in LDS (ds_add_rtn: the slow LDS atomic
// unit, but only ~1.3 operations per thread), reserves each touched list's range with ONE global atomic per
// tile - all of them in flight together - and the threads then write their entries.
// Earlier forms of this pass (measured on the extractor call of BASELINE configs[2]):
//   * one wave-aggregated global atomic per list, counters packed: 124-486 us (32 lists per L2 line);
//   * counters one line apart, the wave looping over its distinct lists with a returning atomic each:
//     24 us - ~8 dependent L2 round trips per wave and ~1000 VALU instructions per thread;
//   * the bounding box of the row's samples instead of the exact tiles: 10 us there, but milliseconds when the
//     samples of a row are far apart (every tile of the box got the row).
constexpr int kCtrStride = 32;      // ints between two list counters (one 128-byte line each)
constexpr int kBinTable = 4096;     // tiles of one level the LDS table covers (more: direct global atomics)

template <typename Src, bool TAPS>
__global__ __launch_bounds__(256) void msda_bin(Src src, TileGeom g, int M, int Lq, int *__restrict__ counter,
                                                int *__restrict__ entries) {
    __shared__ int s_cnt[kBinTable];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < Lq;
    const int y = blockIdx.y;
    const int l = y % g.L, m = (y / g.L) % M, n = y / (g.L * M);
    const int H = g.H[l], W = g.W[l], ntx = g.ntx[l], cap = g.cap[l];
    const int nt = ntx * ((H + kTH - 1) / kTH);
    const bool table = nt <= kBinTable;
    if (table)
        for (int i = threadIdx.x; i < nt; i += 256) s_cnt[i] = 0;
    int tile[4 * kP], rank[4 * kP];
#pragma unroll
    for (int c = 0; c < 4 * kP; ++c) tile[c] = -1, rank[c] = 0;
    if (live) {
        const int64_t row = ((int64_t)n * Lq + q) * M + m;
        const typename Src::Raw raw = src.template load<false>(row, q, l);
        unsigned orphan = 0;
        int ya[kP], yb[kP], xa[kP], xb[kP];
        bool in[kP];
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const float2 xy = src.xy(raw, p, H, W);
            const Base b = make_base(xy.x, xy.y, H, W);
            // tile rows / columns of the corner rows y0, y0 + 1 clipped to the map (inside: -1 <= y0 <= H - 1)
            ya[p] = max(b.y0, 0) >> kTShY, yb[p] = min(b.y0 + 1, H - 1) >> kTShY;
            xa[p] = max(b.x0, 0) >> kTShX, xb[p] = min(b.x0 + 1, W - 1) >> kTShX;
            in[p] = b.inside;
            orphan |= b.inside ? 0u : 1u << p;          // no corner anywhere: no tile will own this sample
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int tyc = (c >> 1) ? yb[p] : ya[p], txc = (c & 1) ? xb[p] : xa[p];
                // the block of the sample itself: second row / column only if it is another tile
                bool fresh = in[p] && ((c >> 1) == 0 || yb[p] != ya[p]) && ((c & 1) == 0 || xb[p] != xa[p]);
#pragma unroll
                for (int e = 0; e < kP; ++e)            // not in the block of an earlier sample
                    if (e < p) fresh = fresh && !(in[e] && tyc >= ya[e] && tyc <= yb[e] && txc >= xa[e] && txc <= xb[e]);
                tile[p * 4 + c] = fresh ? tyc * ntx + txc : -1;
            }
        }
        // a sample inside the gate has its corner (max(y0,0), max(x0,0)) in the map, so some tile owns it and writes
        // its gradients; the others (gate failed: spec cuh:288) get their zeros here
        if (TAPS && orphan) {
            const float z[kP] = {0.f, 0.f, 0.f, 0.f};
            src.store_grads(row, l, orphan, z, z, z, H, W);
        }
    }
    const int64_t list0 = ((int64_t)n * g.T + g.tbase[l]) * M + m;         // + tile * M
    const int64_t ent0 = ((int64_t)n * g.ET + g.ebase[l]) * M + (int64_t)m * cap;      // + tile * cap * M
    __syncthreads();
    if (table) {
#pragma unroll
        for (int c = 0; c < 4 * kP; ++c)
            if (tile[c] >= 0) rank[c] = atomicAdd(&s_cnt[tile[c]], 1);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += 256) {
            const int c = s_cnt[i];
            if (c) s_cnt[i] = atomicAdd(counter + (list0 + (int64_t)i * M) * kCtrStride, c);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4 * kP; ++c)
            if (tile[c] >= 0) {
                const int pos = s_cnt[tile[c]] + rank[c];
                if (pos < cap) entries[ent0 + (int64_t)tile[c] * cap * M + pos] = q;
            }
    } else {
#pragma unroll
        for (int c = 0; c < 4 * kP; ++c)
            if (tile[c] >= 0) {
                const int pos = atomicAdd(counter + (list0 + (int64_t)tile[c] * M) * kCtrStride, 1);
                if (pos < cap) entries[ent0 + (int64_t)tile[c] * cap * M + pos] = q;
            }
    }
}

// ---- pass 4: one workgroup per list ---------------------------------------------------------------
// LDS per wave: Wt[32 px + 1 dummy][WS] fp32 + G[64 entries][32 ch] in the grad_out dtype.
//   bf16: WS = 68 (272-byte rows: the 8-float fragment reads are conflict-free ds_read_b128)
//   fp32: WS = 65 (the one-float fragment reads of v_mfma_f32_32x32x2_f32 are conflict-free)
constexpr int kWinW = kTW + 1, kWinH = kTH + 1;        // value window of a tile: the tile + one pixel to the right / below

template <typename GT>
struct TileLds {
    static constexpr int WS = std::is_same<GT, float>::value ? 65 : 68;
    static constexpr int w_floats = ((kTilePx + 1) * WS + 3) / 4 * 4;      // + one dummy row for the corners outside the tile
    static constexpr int g_bytes = 64 * kD * (int)sizeof(GT);
    static constexpr int v_bytes = kWinW * kWinH * kD * (int)sizeof(GT);   // value rows of the window (16-byte multiple)
    static constexpr int per_wave_taps = (w_floats * 4 + g_bytes + v_bytes + 15) / 16 * 16;
    static constexpr int per_wave_plain = (w_floats * 4 + g_bytes + 15) / 16 * 16;
};

// <grad_out row, value row> over the 32 channels; g: the lane's grad_out row as loaded (raw 16-byte pieces)
template <typename GT>
__device__ __forceinline__ float row_dot(const uint4 (&g)[kD * sizeof(GT) / 16], const unsigned char *vrow) {
    float d = 0.f;
    if constexpr (sizeof(GT) == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = *reinterpret_cast<const uint4 *>(vrow + 16 * i);
            d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, g[i].x), __builtin_bit_cast(bf16x2, v.x), d, false);
            d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, g[i].y), __builtin_bit_cast(bf16x2, v.y), d, false);
            d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, g[i].z), __builtin_bit_cast(bf16x2, v.z), d, false);
            d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, g[i].w), __builtin_bit_cast(bf16x2, v.w), d, false);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(vrow + 16 * i);
            d += __builtin_bit_cast(float, g[i].x) * v.x + __builtin_bit_cast(float, g[i].y) * v.y +
                 __builtin_bit_cast(float, g[i].z) * v.z + __builtin_bit_cast(float, g[i].w) * v.w;
        }
    }
    return d;
}

// TAPS: the tile also computes d(location) / d(attention) of the samples it owns (else a gather kernel of
// msda.hip / msda_fused.hip does, and the value window is not loaded).
template <typename GT, typename OT, typename Src, bool TAPS>
__global__ __launch_bounds__(kTileThreads) void msda_tile_gv(Src src, TileGeom g, int M, int64_t Lq, int64_t S,
                                                             const GT *__restrict__ value, const GT *__restrict__ grad_out,
                                                             int *__restrict__ counter,
                                                             const int *__restrict__ entries, int64_t nwgs, int ablate,
                                                             float *__restrict__ slabs, OT *__restrict__ grad_value) {
    using LD = TileLds<GT>;
    constexpr int WS = LD::WS;
    constexpr bool F32 = std::is_same<GT, float>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    float *Wt = reinterpret_cast<float *>(smem);                 // one wave per workgroup
    GT *Gs = reinterpret_cast<GT *>(smem + (size_t)LD::w_floats * 4);
    unsigned char *Vs = smem + (size_t)LD::w_floats * 4 + LD::g_bytes;

    const int64_t wg = xcd_chunked_block(nwgs);
    if (wg >= nwgs) return;
    // workgroup -> (level, n, tile, head m, slice j): the slices and heads of one tile are neighbours in the
    // launch order (they read the same query rows: L2 hits instead of 64-byte pieces of lines from HBM)
    int blk = 0;
#pragma unroll
    for (int i = 1; i < kMaxL; ++i) blk = (i < g.L && wg >= g.wgbase[i]) ? i : blk;
    const int l = g.wgorder[blk];
    const int ks = g.ksplit[l];
    int64_t r_ = wg - g.wgbase[blk];
    const int j = (int)(r_ % ks);
    r_ /= ks;
    const int m = (int)(r_ % M);
    r_ /= M;
    const int tl = (int)(r_ % g.ntiles[l]);
    const int64_t n = r_ / g.ntiles[l];
    const int tile_all = g.tbase[l] + tl;
    const int64_t lst = ((int64_t)n * g.T + tile_all) * M + m;
    const int H = g.H[l], W = g.W[l], ntx = g.ntx[l];
    const int ty = tl / ntx, tx = tl - ty * ntx;

    // zero this wave's Wt (16-byte stores; the region is a multiple of 16 bytes)
    for (int i = lane * 4; i < LD::w_floats; i += 64 * 4)
        *reinterpret_cast<float4 *>(Wt + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    // the value rows of this head in the tile's window (tile + 1 pixel right / below; zeros outside the map):
    // what the samples OWNED by this tile interpolate for d(location) / d(attention)
    constexpr int ROWB = kD * (int)sizeof(GT);
    if (TAPS) {
        constexpr int PCS = ROWB / 16;
        const int64_t vstride = (int64_t)M * kD;
        const GT *vmap = value + ((int64_t)n * S + g.start[l]) * vstride + m * kD;
        for (int i = lane; i < kWinW * kWinH * PCS; i += 64) {
            const int wp = i / PCS, pc = i - wp * PCS;
            const int yy = ty * kTH + wp / kWinW, xx = tx * kTW + wp % kWinW;
            const bool in = yy < H && xx < W;
            const uint4 v = *reinterpret_cast<const uint4 *>(
                reinterpret_cast<const unsigned char *>(vmap + ((int64_t)min(yy, H - 1) * W + min(xx, W - 1)) * vstride) + 16 * pc);
            *reinterpret_cast<uint4 *>(Vs + wp * ROWB + 16 * pc) = in ? v : make_uint4(0, 0, 0, 0);
        }
    }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // the list, or - if it overflowed its capacity - every query of this (n, head, level).  The first 64
    // entry slots are requested together with the counter (every list has >= 128 slots): one round trip less
    // in front of the first chunk, which is most of a short list's life.
    const int *list = entries + list_entry_base(g, n, l, tile_all, M, m);
    const int first_slot = list[lane];
    const int count = counter[lst * kCtrStride];
    const bool scan_all = count > g.cap[l];
    const int nent = scan_all ? (int)Lq : count;
    const int nchunks = (nent + 63) / 64;
    // slice j of the list's ks workgroups takes chunks j, j + ks, ...; only min(ks, nchunks) of them have any
    // (at least one - slice 0 - so that an empty tile is still stored)
    const int nslices = min(ks, max(nchunks, 1));
    if (j >= nslices) return;
    if ((ablate & 32) && count >= 0) return;                    // timing only: workgroup launch + its first two loads
    constexpr int NV = kD * (int)sizeof(GT) / 16;               // 16-byte pieces of a grad_out row
    const int STEP = ks;
    // Software pipeline, one iteration deep for the operands and two for the list entries: at the top of the
    // iteration of chunk i the wave issues the loads of chunk i + 1's samples and grad_out row (its entry
    // index was requested an iteration earlier) and of chunk i + 2's entry index, then works on chunk i,
    // whose operands were requested an iteration ago.  Every load is unconditional: lanes past the end of the
    // list re-read the list's last entry and are masked afterwards (a load under a condition is waited on
    // alone).
    auto entry_of = [&](int chunk) -> int64_t {
        const int i = min(chunk * 64 + lane, nent - 1);
        return scan_all ? i : list[i];
    };
    auto first_entry = [&]() -> int64_t {                       // chunk j
        if (j != 0) return entry_of(j);
        return scan_all ? lane : (lane < nent ? first_slot : 0);        // slots past the count hold garbage
    };
    auto live_of = [&](int chunk) -> bool { return chunk < nchunks && chunk * 64 + lane < nent; };
    bool live = false;
    int64_t row = 0;                                            // (n, q, m) row of this lane's entry
    typename Src::Raw raw{};
    uint4 gr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) gr[i] = make_uint4(0, 0, 0, 0);
    int64_t q_n = 0;                                            // entry of the wave's next chunk
    if (j < nchunks) {
        const int64_t q = first_entry();
        q_n = entry_of(j + STEP);
        live = live_of(j);
        row = (n * Lq + q) * M + m;
        raw = src.template load<true>(row, q, l);
        const uint4 *src4 = reinterpret_cast<const uint4 *>(grad_out + row * kD);
#pragma unroll
        for (int i = 0; i < NV; ++i) gr[i] = src4[i];
    }
    for (int ch = j; ch < ((ablate & 64) ? 0 : nchunks); ch += STEP) {
        // ---- requests for the chunks to come
        const bool live_n = live_of(ch + STEP);
        const int64_t row_n = (n * Lq + q_n) * M + m;
        const typename Src::Raw raw_n = src.template load<true>(row_n, q_n, l);
        uint4 gr_n[NV];
        {
            const uint4 *src4 = reinterpret_cast<const uint4 *>(grad_out + row_n * kD);
#pragma unroll
            for (int i = 0; i < NV; ++i) gr_n[i] = src4[i];
        }
        const int64_t q_nn = entry_of(ch + 2 * STEP);
        // ---- the entry's grad_out row -> Gs[lane][0..31] (zeros for the lanes past the list's end)
        if (!(ablate & 8)) {
            uint4 *dst4 = reinterpret_cast<uint4 *>(Gs + lane * kD);
#pragma unroll
            for (int i = 0; i < NV; ++i) dst4[i] = live ? gr[i] : make_uint4(0, 0, 0, 0);
        }
        // ---- per sample of the entry:
        //  (1) column `lane` of Wt: attention x bilinear weight of every corner that lands in this tile.  The four
        //      corners of a sample are four different pixels: their read-add-writes go out together (four reads,
        //      then four writes); a corner that is outside the tile / the map / the list goes to the dummy row
        //      with weight 0, so nothing here branches;
        //  (2) if this tile OWNS the sample (its first corner inside the map, in the order 00 01 10 11, lies in the
        //      tile - every sample that passes the gate has exactly one owner, and the owner's list holds it):
        //      d(out)/d(location), d(out)/d(attention) from the value window (spec cuh:126-158: the four corner
        //      dot products <grad_out, value>, each a lane-local chain - no cross-lane reduction).
        float *wcell[kP][4];                     // the cells written, for the clean-up after the matrix phase
        {
            float a[kP], gx[kP], gy[kP], gav[kP];
            unsigned own = 0;
            src.weights(raw, l, a);
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                const float2 xy = src.xy(raw, p, H, W);
                const Base b = make_base(xy.x, xy.y, H, W);
                const float hh = 1.f - b.lh, hw = 1.f - b.lw;
                const float cw[4] = {hh * hw, hh * b.lw, b.lh * hw, b.lh * b.lw};
                const int ry = b.y0 - ty * kTH, rx = b.x0 - tx * kTW;
                const bool on = live && b.inside;
                const bool vy[2] = {on && b.y0 >= 0, on && b.y0 + 1 <= H - 1};       // corner row in the map
                const bool vx[2] = {b.x0 >= 0, b.x0 + 1 <= W - 1};
                const bool iy[2] = {(unsigned)ry < (unsigned)kTH, (unsigned)(ry + 1) < (unsigned)kTH};     // ... in the tile
                const bool ix[2] = {(unsigned)rx < (unsigned)kTW, (unsigned)(rx + 1) < (unsigned)kTW};
                bool valid[4], mine[4];
                float wgt[4], old[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    valid[c] = vy[c >> 1] && vx[c & 1];
                    mine[c] = valid[c] && iy[c >> 1] && ix[c & 1];
                    const int px = mine[c] ? (ry + (c >> 1)) * kTW + rx + (c & 1) : kTilePx;
                    wcell[p][c] = Wt + px * WS + lane;
                    wgt[c] = mine[c] ? a[p] * cw[c] : 0.f;
                }
                if (!(ablate & 1)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) old[c] = *wcell[p][c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) *wcell[p][c] = old[c] + wgt[c];
                }
                const bool owned = valid[0] ? mine[0] : valid[1] ? mine[1] : valid[2] ? mine[2] : (valid[3] && mine[3]);
                own |= owned ? 1u << p : 0u;
                gx[p] = gy[p] = gav[p] = 0.f;
                if (TAPS && !(ablate & 4) && __ballot(owned)) {                              // wave-uniform: some lane owns its sample p
                    float d[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int wy = min(max(ry + (c >> 1), 0), kWinH - 1), wx = min(max(rx + (c & 1), 0), kWinW - 1);
                        const float dc = row_dot<GT>(gr, Vs + (wy * kWinW + wx) * ROWB);
                        d[c] = valid[c] ? dc : 0.f;
                    }
                    gav[p] = cw[0] * d[0] + cw[1] * d[1] + cw[2] * d[2] + cw[3] * d[3];
                    gy[p] = (hw * (d[2] - d[0]) + b.lw * (d[3] - d[1])) * a[p];
                    gx[p] = (hh * (d[1] - d[0]) + b.lh * (d[3] - d[2])) * a[p];
                }
            }
            if (TAPS && !(ablate & 16) && own) src.store_grads(row, l, own, gx, gy, gav, H, W);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- dV[32 px, 32 ch] += Wt[32 px, 64 k] x G[64 k, 32 ch]
        if (ablate & 2) {
        } else if constexpr (F32) {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll 8
            for (int kk = 0; kk < 32; ++kk) {
                const int k = 2 * kk + h;
                const float bfr = reinterpret_cast<const float *>(Gs)[k * kD + r];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wt[r * WS + k], bfr, acc, 0, 0, 0);
            }
        } else {
            const int r = lane & 31, h = lane >> 5;
            const int grp = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
            // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group names row q, columns 4p..4p+3 of a 4 x 16 block
            // and receives column (lane & 15) of its 4 rows (checked on the GPU: tools/ubench/tr_read.hip).
            // Group g: columns 16(g&1).., rows 8(g>>1) + 4r + q.
            const __bf16 *gbase = reinterpret_cast<const __bf16 *>(Gs) + (8 * (grp >> 1) + qq) * kD + 16 * (grp & 1) + 4 * pp;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(gbase + (16 * s) * kD));
                const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(gbase + (16 * s + 4) * kD));
                // whole-vector reinterpretation (an element-wise short -> __bf16 bit_cast was compiled into a
                // fragment that repeated one dword of each read)
                const s16x8 t01 = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 bfr = __builtin_bit_cast(bf16x8, t01);
                const float *wp = Wt + r * WS + 16 * s + 8 * h;
                const float4 w0 = *reinterpret_cast<const float4 *>(wp);
                const float4 w1 = *reinterpret_cast<const float4 *>(wp + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                bf16x8 ahi, alo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ahi[j] = (__bf16)wv[j];
                    alo[j] = (__bf16)(wv[j] - (float)ahi[j]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bfr, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bfr, acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- put the column back to zero (what this lane wrote; the dummy row may hold anything)
        if (!(ablate & 1))
#pragma unroll
        for (int p = 0; p < kP; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) *wcell[p][c] = 0.f;
        __builtin_amdgcn_wave_barrier();
        live = live_n, raw = raw_n, q_n = q_nn, row = row_n;
#pragma unroll
        for (int i = 0; i < NV; ++i) gr[i] = gr_n[i];
    }

    // ---- the tile: accumulator -> LDS (row-major) -> whole rows, 16 bytes per lane.  A list shared by several
    // workgroups goes through per-slice slabs in the workspace: every slice stores its partial tile, the slice
    // that arrives last adds them up in slice order - so the sum does not depend on who arrives when - and
    // stores the tile.
    static_assert(kTileThreads == 64, "one wave per workgroup");
    float *red = reinterpret_cast<float *>(smem);               // [32 px][32 ch] fp32 (Wt is no longer needed)
    {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((i & 3) + 8 * (i >> 2) + 4 * h) * kD + r] = acc[i];
    }
    __builtin_amdgcn_wave_barrier();
    // Hand-off without cache-wide fences (a release / acquire pair per workgroup writes back and invalidates a
    // whole L2 / L1: with thousands of lists that was 5x the kernel): every slab store is a write-through
    // (`sc1`: relaxed agent-scope atomic store), the wave drains them (s_waitcnt vmcnt(0)) before ONE lane adds to
    // the list's arrival counter, and the wave whose add comes last reads the slabs back with `sc1` loads (relaxed
    // agent-scope atomic loads: served by L2, never by a stale L1) - the measured-valid form of
    // MI355X_MICROARCH.md "Valid forms", first row, for single-wave workgroups.
    const float *part = nullptr;                                // slabs of the list, if it is shared
    if (nslices > 1) {
        float *mine = slabs + (g.slab[l] + (((int64_t)n * g.ntiles[l] + tl) * M + m) * ks) * (kTilePx * kD);
        {
            const int r = lane & 31, h = lane >> 5;
            float *dst = mine + (int64_t)j * kTilePx * kD + r;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __hip_atomic_store(dst + ((i & 3) + 8 * (i >> 2) + 4 * h) * kD, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(counter + lst * kCtrStride + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived != nslices - 1) return;
        part = mine;
    }
    const int64_t stride = (int64_t)M * kD;
    OT *gv = grad_value + (n * S + g.start[l]) * stride + m * kD;
    constexpr int CH = 16 / (int)sizeof(OT);                    // channels per 16-byte piece
    constexpr int PIECES = kTilePx * kD / CH;
    for (int i = lane; i < PIECES; i += 64) {
        const int px = i / (kD / CH), c0 = (i % (kD / CH)) * CH;
        const int yy = ty * kTH + (px >> kTShX), xx = tx * kTW + (px & (kTW - 1));
        if (yy >= H || xx >= W) continue;
        float v[CH];
        if (part) {
#pragma unroll
            for (int c = 0; c < CH; ++c) v[c] = 0.f;
            for (int sl = 0; sl < nslices; ++sl)
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    v[c] += __hip_atomic_load(part + (int64_t)sl * kTilePx * kD + px * kD + c0 + c, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int c = 0; c < CH; c += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(red + px * kD + c0 + c);
                v[c] = a.x, v[c + 1] = a.y, v[c + 2] = a.z, v[c + 3] = a.w;
            }
        }
        OT *dst = gv + ((int64_t)yy * W + xx) * stride + c0;
        if constexpr (std::is_same<OT, float>::value) {
            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            bf16x8 o;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = (__bf16)v[c];
            *reinterpret_cast<bf16x8 *>(dst) = o;
        }
    }
}

// ---- fused core only: softmax backward of the attention logits --------------------------------------
// d_logit[s] = p_s * (ga_s - sum_t p_t ga_t), p = softmax(logits of the (n, q, m) row), ga = d(out)/d(p) as the
// tile pass (and, for gated samples, the binning pass) left it in the workspace.  One thread per row.
template <typename PT, int L>
__global__ __launch_bounds__(256) void msda_logit_grad(const PT *__restrict__ logit, const float *__restrict__ ga, int64_t rows,
                                                       PT *__restrict__ d_logit) {
    constexpr int LP = L * kP;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float p[LP], g[LP];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < LP; ++s) {
        p[s] = (float)logit[row * LP + s];
        mx = fmaxf(mx, p[s]);
    }
#pragma unroll
    for (int s = 0; s < LP; s += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(ga + row * LP + s);
        g[s] = v.x, g[s + 1] = v.y, g[s + 2] = v.z, g[s + 3] = v.w;
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < LP; ++s) {
        p[s] = __expf(p[s] - mx);
        sum += p[s];
    }
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int s = 0; s < LP; ++s) {
        p[s] *= inv;
        dot += p[s] * g[s];
    }
#pragma unroll
    for (int s = 0; s < LP; ++s) d_logit[row * LP + s] = (PT)(p[s] * (g[s] - dot));
}

// ---- host side -----------------------------------------------------------------------------------
struct Plan {
    TileGeom g;
    int64_t nlists;                          // N * T * M
    int64_t nwgs;                            // workgroups of the tile pass
    int64_t off_counts, off_entries, off_slabs, off_ga, total;      // byte offsets in the workspace
};

constexpr int kChunksPerWg = 8;              // a list longer than this many 64-entry chunks (by capacity) is shared

// Levels the tile path takes: every level a proper window of [0, S) (the kernels of msda.hip skip a level that
// fails this guard; here the caller falls back to them).
int make_plan(const char *fn, int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq, int64_t P,
              const int64_t *shapes_host, const int64_t *lsi_host, Plan *pl) {
    if (L < 1 || L > kMaxL || P != kP || !shapes_host || !lsi_host)
        return fail(VAH_E_UNSUPPORTED, "%s: the tiled path needs 1 <= L <= %d, P == %d and the host copy of the level geometry",
                    fn, kMaxL, kP);
    if (Lq >= (1 << 24) || S >= ((int64_t)1 << 31) || N * M * L >= 65536)
        return fail(VAH_E_UNSUPPORTED, "%s: problem too large for the tiled path", fn);
    TileGeom g{};
    g.L = (int)L;
    int64_t T = 0, ET = 0, nwgs = 0, nslabs = 0, even_load[kMaxL] = {0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
        const int64_t H = shapes_host[2 * l], W = shapes_host[2 * l + 1], st = lsi_host[l];
        if (H < 1 || W < 1 || st < 0 || st + H * W > S || H > 32760 || W > 32760)
            return fail(VAH_E_UNSUPPORTED, "%s: level %d (%lld x %lld at %lld) is not a window of [0, S)", fn, l, (long long)H,
                        (long long)W, (long long)st);
        g.H[l] = (int)H, g.W[l] = (int)W, g.start[l] = (int)st;
        g.ntx[l] = (int)((W + kTW - 1) / kTW);
        const int64_t nt = (int64_t)g.ntx[l] * ((H + kTH - 1) / kTH);
        g.tbase[l] = (int)T;
        // capacity: 12 rows' worth per tile of evenly spread queries (a (q, level) row names ~1.5 tiles when its
        // samples sit close together - the adapter - and ~5.6 when each sample falls somewhere else - test.py's
        // uniform recipe: 2x headroom over that)
        int64_t cap = (12 * Lq + nt - 1) / nt;
        cap = (cap < 128 ? 128 : cap + 63) / 64 * 64;
        if (cap > Lq + 64) cap = (Lq + 63) / 64 * 64;          // a list never holds more than every query
        g.cap[l] = (int)cap;
        g.ebase[l] = (int)ET;
        g.ntiles[l] = (int)nt;
        // shared lists: by the even load (2 tiles per row), not by the capacity - the adapter's lists (a few
        // chunks) stay with one workgroup, the coarse levels of a pyramid (every query in each of a few tiles) do not
        const int64_t even = (2 * Lq + nt - 1) / nt < cap ? (2 * Lq + nt - 1) / nt : cap;
        g.ksplit[l] = (int)((even / 64 + kChunksPerWg) / kChunksPerWg);
        even_load[l] = even;
        g.slab[l] = nslabs;
        if (g.ksplit[l] > 1) nslabs += N * nt * M * g.ksplit[l];
        T += nt;
        ET += nt * cap;
        if (T >= (1 << 24) || ET >= ((int64_t)1 << 30)) return fail(VAH_E_UNSUPPORTED, "%s: too many tiles", fn);
    }
    const int64_t nlists = N * M * T;
    if (nlists >= ((int64_t)1 << 26) || N * M * ET >= ((int64_t)1 << 33))
        return fail(VAH_E_UNSUPPORTED, "%s: problem too large for the tiled path", fn);
    g.T = (int)T;
    g.ET = (int)ET;
    pl->g = g;
    pl->nlists = nlists;
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    // launch order: the levels with the longest lists first (their workgroups run longest: no tail of a few
    // long lists at the end of the launch).  wgorder[i] = level of the i-th block of workgroups.
    int order[kMaxL];
    for (int l = 0; l < L; ++l) order[l] = l;
    for (int a = 1; a < L; ++a)
        for (int b = a; b > 0 && even_load[order[b]] / g.ksplit[order[b]] > even_load[order[b - 1]] / g.ksplit[order[b - 1]]; --b) {
            const int t = order[b];
            order[b] = order[b - 1];
            order[b - 1] = t;
        }
    for (int i = 0; i < L; ++i) {
        g.wgorder[i] = order[i];
        g.wgbase[i] = nwgs;
        nwgs += N * g.ntiles[order[i]] * M * g.ksplit[order[i]];
    }
    for (int i = (int)L; i <= kMaxL; ++i) g.wgbase[i] = nwgs;
    if (nwgs >= ((int64_t)1 << 31) - 8 || nslabs * kTilePx * kD * 4 >= ((int64_t)1 << 36))
        return fail(VAH_E_UNSUPPORTED, "%s: problem too large for the tiled path", fn);
    pl->g = g;
    pl->nwgs = nwgs;
    pl->off_counts = 0;
    pl->off_entries = up(nlists * kCtrStride * 4);
    pl->off_slabs = pl->off_entries + up(N * M * ET * 4);
    pl->off_ga = pl->off_slabs + up(nslabs * kTilePx * kD * 4);
    pl->total = pl->off_ga + up(N * Lq * M * L * kP * 4);           // fused core: d(out)/d(attention probability), fp32
    return VAH_OK;
}

template <typename GT, typename OT, typename Src, bool TAPS>
int run_tiled(const char *fn, const Src &src, const Plan &pl, int64_t N, int64_t M, int64_t Lq, int64_t S, const GT *value,
              const GT *grad_out, OT *grad_value, void *ws, hipStream_t st) {
    char *base = (char *)ws;
    int *counts = (int *)(base + pl.off_counts), *entries = (int *)(base + pl.off_entries);
    if (hipMemsetAsync(counts, 0, (size_t)pl.nlists * kCtrStride * 4, st) != hipSuccess)
        return fail(VAH_E_SHAPE, "%s: memset failed", fn);
    const dim3 bgrid((unsigned)((Lq + 255) / 256), (unsigned)(N * M * pl.g.L));
    hipLaunchKernelGGL((msda_bin<Src, TAPS>), bgrid, dim3(256), 0, st, src, pl.g, (int)M, (int)Lq, counts, entries);
    if (int rc = check_launch(fn)) return rc;
    const int smem = TAPS ? TileLds<GT>::per_wave_taps : TileLds<GT>::per_wave_plain;
    if (int rc = allow_dynamic_lds((const void *)msda_tile_gv<GT, OT, Src, TAPS>, smem, fn)) return rc;
    const int64_t grid = (pl.nwgs + 7) / 8 * 8;
    // VAH_TILE_ABLATE (timing experiments only, results are wrong): bit 0 skips the Wt updates, 1 the matrix phase, 2 the
    // owned samples' dot products, 3 the grad_out staging, 4 the gradient stores
    static const int ablate = [] {
        const char *e = getenv("VAH_TILE_ABLATE");
        return e ? atoi(e) : 0;
    }();
    hipLaunchKernelGGL((msda_tile_gv<GT, OT, Src, TAPS>), dim3((unsigned)grid), dim3(kTileThreads), smem, st, src, pl.g, (int)M, Lq, S,
                       value, grad_out, counts, (const int *)entries, pl.nwgs, ablate, (float *)(base + pl.off_slabs), grad_value);
    return check_launch(fn);
}

// Who computes d(offsets) / d(logits)?  Measured on BASELINE configs[2] (bf16, profiles/r02_msda_tile_steps.txt):
// one level of many-entry lists (extractor: 64x64 map, ~240 entries per tile): inside the tile pass 108 us per call
// against 127 us with the gather kernel of msda_fused.hip in front; three levels of short lists (injector):
// 172 us against 160 us.  fp32 values (32 FMAs per corner instead of 16 packed dot products, 250 registers):
// always the gather kernel.
template <typename VT, int L>
constexpr bool kTapsInTile = std::is_same<VT, __bf16>::value && L == 1;

template <typename VT, typename PT, int L>
int fused_tiled(const char *fn, const Plan &pl, const void *value, const void *off, const void *logit, const float *ref,
                int ref_levels, int64_t N, int64_t M, int64_t Lq, int64_t S, const void *grad_out, void *grad_value, int gv_bf16,
                void *d_off, void *d_logit, void *ws, hipStream_t st) {
    constexpr bool TAPS = kTapsInTile<VT, L>;
    float *ga = (float *)((char *)ws + pl.off_ga);
    FusedSrc<PT, L> src{(const PT *)off, (const PT *)logit, ref, ref_levels, (PT *)d_off, ga};
    int rc;
    if (gv_bf16) {
        if constexpr (std::is_same<VT, __bf16>::value)
            rc = run_tiled<VT, __bf16, FusedSrc<PT, L>, TAPS>(fn, src, pl, N, M, Lq, S, (const VT *)value, (const VT *)grad_out,
                                                            (__bf16 *)grad_value, ws, st);
        else
            return fail(VAH_E_UNSUPPORTED, "%s: a bf16 grad_value needs bf16 values", fn);
    } else {
        rc = run_tiled<VT, float, FusedSrc<PT, L>, TAPS>(fn, src, pl, N, M, Lq, S, (const VT *)value, (const VT *)grad_out,
                                                       (float *)grad_value, ws, st);
    }
    if (rc || !TAPS) return rc;
    const int64_t rows = N * Lq * M;
    hipLaunchKernelGGL((msda_logit_grad<PT, L>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (const PT *)logit,
                       (const float *)ga, rows, (PT *)d_logit);
    return check_launch(fn);
}

}  // namespace
}  // namespace vah

extern "C" {

int64_t vah_msda_tile_ws_bytes(int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq, int64_t P,
                               const int64_t *shapes_host, const int64_t *lsi_host) {
    vah::clear_error();
    vah::Plan pl;
    if (vah::make_plan("vah_msda_tile_ws_bytes", N, S, M, L, Lq, P, shapes_host, lsi_host, &pl)) return -1;
    return pl.total;
}

int vah_msda_backward_tiled_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                                const float *attn, const float *grad_out, int64_t N, int64_t S, int64_t M, int64_t D,
                                int64_t L, int64_t Lq, int64_t P, float *grad_value, float *grad_loc, float *grad_attn,
                                const int64_t *shapes_host, const int64_t *lsi_host, void *ws, int64_t ws_bytes,
                                void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_backward_tiled_f32";
    if (N < 0 || S < 1 || M < 1 || Lq < 0 || M * D >= (1LL << 31)) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (D != kD) return fail(VAH_E_UNSUPPORTED, "%s: needs D == 32", fn);
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn || !ws)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)grad_out | (uintptr_t)grad_value | (uintptr_t)ws) % 16 || ((uintptr_t)loc | (uintptr_t)grad_loc) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    Plan pl;
    if (int rc = make_plan(fn, N, S, M, L, Lq, P, shapes_host, lsi_host, &pl)) return rc;
    if (ws_bytes < pl.total) return fail(VAH_E_SHAPE, "%s: workspace too small (%lld < %lld)", fn, (long long)ws_bytes, (long long)pl.total);
    hipStream_t st = (hipStream_t)stream;
    // SURVEY.md 8d bytes of the fp32 backward
    LaunchScope scope("msda_bwd_f32", 4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D), st);
    // d(loc), d(attn): the 8-lane gather kernel of msda.hip (fp32: cheaper than inside the tile pass, see kTapsInTile)
    if (int rc = msda_grad_taps_f32(value, shapes, lsi, loc, attn, grad_out, N, S, M, D, L, Lq, P, grad_loc, grad_attn, st))
        return rc;
    PlainSrc src{loc, attn, (int)(L * P), grad_loc, grad_attn};
    return run_tiled<float, float, PlainSrc, false>(fn, src, pl, N, M, Lq, S, value, grad_out, grad_value, ws, st);
}

int vah_msda_fused_backward_tiled(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                                  const void *offsets, const void *logits, int param_dtype, const float *ref,
                                  int64_t ref_levels, const void *grad_out, int64_t N, int64_t S, int64_t M, int64_t D,
                                  int64_t L, int64_t Lq, int64_t P, void *grad_value, int grad_value_dtype,
                                  void *d_offsets, void *d_logits, const int64_t *shapes_host, const int64_t *lsi_host,
                                  void *ws, int64_t ws_bytes, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_fused_backward_tiled";
    if (N < 0 || S < 1 || M < 1 || Lq < 0 || M * D >= (1LL << 31)) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (D != kD) return fail(VAH_E_UNSUPPORTED, "%s: needs D == 32", fn);
    if (ref_levels != 1 && ref_levels != L) return fail(VAH_E_SHAPE, "%s: ref_levels must be 1 or L", fn);
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !offsets || !logits || !ref || !grad_out || !grad_value || !d_offsets || !d_logits || !ws)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)grad_out | (uintptr_t)grad_value | (uintptr_t)ws) % 16 || ((uintptr_t)offsets | (uintptr_t)ref) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    if ((value_dtype | param_dtype | grad_value_dtype) & ~1) return fail(VAH_E_UNSUPPORTED, "%s: dtype codes must be 0 (f32) or 1 (bf16)", fn);
    Plan pl;
    if (int rc = make_plan(fn, N, S, M, L, Lq, P, shapes_host, lsi_host, &pl)) return rc;
    if (ws_bytes < pl.total) return fail(VAH_E_SHAPE, "%s: workspace too small (%lld < %lld)", fn, (long long)ws_bytes, (long long)pl.total);
    hipStream_t st = (hipStream_t)stream;
    const int64_t vs = value_dtype ? 2 : 4, ps = param_dtype ? 2 : 4, gs = grad_value_dtype ? 2 : 4;
    LaunchScope scope("msda_fused_bwd", vs * (N * S * M * D + N * Lq * M * D) + gs * N * S * M * D + ps * 6 * N * Lq * M * L * P, st,
                      4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D));
    // d(offsets), d(logits) from the gather kernel of msda_fused.hip (nothing scattered) where the tile pass does not
    // compute them itself
    if (!(value_dtype == 1 && L == 1))
        if (int rc = msda_fused_grad_taps(value, value_dtype, shapes, lsi, offsets, logits, param_dtype, ref, ref_levels, grad_out,
                                          N, S, M, L, Lq, P, d_offsets, d_logits, st))
            return rc;
#define VAH_CASE(VT, VC, PT, PC, LL)                                                                                     \
    if (value_dtype == VC && param_dtype == PC && L == LL)                                                               \
        return fused_tiled<VT, PT, LL>(fn, pl, value, offsets, logits, ref, (int)ref_levels, N, M, Lq, S, grad_out,      \
                                       grad_value, grad_value_dtype, d_offsets, d_logits, ws, st)
#define VAH_CASES(LL)                       \
    VAH_CASE(float, 0, float, 0, LL);       \
    VAH_CASE(__bf16, 1, __bf16, 1, LL);     \
    VAH_CASE(__bf16, 1, float, 0, LL);      \
    VAH_CASE(float, 0, __bf16, 1, LL)
    VAH_CASES(1);
    VAH_CASES(3);
    VAH_CASES(4);
#undef VAH_CASES
#undef VAH_CASE
    return fail(VAH_E_UNSUPPORTED, "%s: L = %lld not instantiated", fn, (long long)L);
}

}  // extern "C"
