// Atomic-free backward of multi-scale deformable attention for gfx950 (round 3 form).
//
// Replaces the scatter of /root/reference/detection/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159
// (ms_deform_attn_col2im_bilinear: 4 atomicAdd per sample and channel, called from :301-403) for ANY
// sampling locations - the plain MSDeformAttnFunction (no reference grid, e.g. the Mask2Former pixel
// decoder with per-batch reference points, seg/.../msdeformattn_pixel_decoder.py:224-242) as well as
// the fused MSDeformAttn core - and, for bf16 operands, the d(loc) / d(attn) sums of cuh:126-158 too.
//
// Nothing here needs the level geometry on the HOST (the reference reads spatial_shapes / level_start_index in
// the kernel, cuh:274-277; so do these): grids and the workspace are sized from upper bounds that follow from
// (N, S, M, L, Lq) alone, the plan (tiles, list capacities, work items) is computed on the device by the first
// kernel.  No D2H read, no host cache: the call can be captured in a HIP graph with fresh shape tensors.
//
// Kernels per call (all on the caller's stream, workspace from the caller):
//   0. msda_plan : level table -> workspace; list counters zeroed (replaces a memset); if the levels do NOT tile
//                  [0, S) exactly (gaps / overlaps / a level outside the value rows) grad_value is zero-filled here
//                  and the tile pass ADDS with atomics instead of storing (the reference's semantics for any
//                  geometry: zeros + atomics).
//   1. msda_bin  : one thread per (n, m, level, q): the 8x4-pixel tiles its P = 4 samples touch as a bit mask over a
//                  window of 4 x 8 tiles anchored at the row's first tile (a row spread wider takes an exact,
//                  slower walk); per-workgroup counting in LDS, ONE returning global atomic per touched list and
//                  workgroup, 4-byte entries (q).  Lists have a fixed capacity per level; a list that overflows is
//                  not used: its tile walks all queries instead (slow, exact).
//   2. msda_tile : PERSISTENT single-wave workgroups (as many as the chip holds at once).  A wave walks its work
//                  items - (n, head, tile) lists, the 12 heads of a tile adjacent and on one XCD - back to back, with
//                  the next item's list head and first operands in flight while it works on the current one (a
//                  one-list-per-workgroup launch spent 6 us of launch + two dependent round trips per list: 32 of
//                  88 us on the injector call, and ran 3072 lists on 2560 slots in two rounds on the extractor).
//                  Per chunk of 64 list entries (one per lane):
//                    * dV^T[32 ch, 32 px] += G^T[32 ch, 64 entries] x Wt^T[64 entries, 32 px] on the matrix cores
//                      (lane k builds column k of Wt - attention x bilinear weight of its entry's corners inside the
//                      tile - with plain LDS read-add-writes on its OWN column; bf16 rows: v_mfma_f32_32x32x16_bf16
//                      with Wt as bf16 hi + lo; fp32 rows: v_mfma_f32_32x32x2_f32).  The accumulator has the pixel
//                      on the lane: the tile is stored straight from registers (8 / 16 bytes per lane and row);
//                    * bf16 rows: the corner dot products <grad_out row, value row> of every entry against the
//                      tile's 10 x 6-pixel value window as ONE more product Dd^T[60 px, 64 entries] = V x G^T (value
//                      fragments stay in registers for the whole list), handed to the entry's lane through LDS;
//                      a sample is OWNED by the tile of its first in-map corner and the owner writes d(offset) and
//                      d(out)/d(attention) (round 2 formed them as 16 x 16 packed dot products per entry on the VALU:
//                      a quarter of the kernel).
//   3. msda_grad_finish (fused core only): per-sample gradient records -> d_offsets; softmax backward -> d_logits.
#include <cstdlib>
#include <type_traits>

#include "msda_common.h"
#include "msda_internal.h"

#ifndef VAH_GR_COOP
#define VAH_GR_COOP 1
#endif
#ifndef VAH_VF_PER_CHUNK
#define VAH_VF_PER_CHUNK 0
#endif
// Timing experiments only (tools/ablate_tile.sh builds a SEPARATE library with -DVAH_TILE_ABLATE=bits; the results of
// such a build are wrong): bit 0 skips the Wt updates, 1 the matrix phase, 2 the owned samples' dot products, 4 the
// gradient stores, 6 the chunk loop.  Compile-time, so the shipped library has neither the branches nor a switch.
#ifndef VAH_TILE_ABLATE
#define VAH_TILE_ABLATE 0
#endif

namespace vah {
namespace {

using namespace vah::msda;

constexpr int ablate = VAH_TILE_ABLATE;

constexpr int kTW = 8, kTH = 4;    // tile: 8 x 4 pixels = the 32 columns of one 32x32 MFMA tile
constexpr int kTShY = 2, kTShX = 3;
constexpr int kTilePx = kTW * kTH;
constexpr int kP = 4;              // points per level on this path
constexpr int kMaxL = 4;
constexpr int kD = 32;
constexpr int kCtrStride = 32;     // ints between two list counters (one 128-byte line each)
constexpr int kBinTable = 2048;    // tiles of one level the LDS tables of the binning pass cover (more: global atomics)
constexpr int kChunksPerWg = 8;    // a list longer than this many 64-entry chunks (by its even load) is shared
constexpr int kWinW = kTW + 2, kWinH = kTH + 2, kWinPx = kWinW * kWinH;   // value window: tile + 1 pixel all around

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;

// ---- the plan: computed on the device, kept at the head of the workspace -----------------------------------------
struct PlanDev {
    int L, T, ET, partition;       // levels; tiles per (n, m); entry slots per (n, head); 1 = the levels tile [0, S) exactly
    int H[kMaxL], W[kMaxL], start[kMaxL], ntx[kMaxL], ntiles[kMaxL], tbase[kMaxL];
    int cap[kMaxL], ebase[kMaxL];  // list capacity of the level; first entry slot of the level's lists (per n and head)
    int valid[kMaxL];              // the level is a window of [0, S)
    // a list of level l is shared by ksplit[l] work items (1 for the adapter's calls; a coarse level - few tiles, every
    // query in each - gets more); items are ordered longest lists first: wgorder[i] = level of the i-th block of items;
    // slab[l] = first partial-sum slab of level l (levels with ksplit 1: none)
    int ksplit[kMaxL], wgorder[kMaxL];
    int slab[kMaxL];
    int xcount[8];                 // items of the tile pass per XCD (see ItemDesc)
    int gb[kMaxL + 1], ib[kMaxL + 1];      // first group / first item (global numbering) of level block b
    int nwgs, nlists;
};
constexpr int kPlanBytes = 512;
static_assert(sizeof(PlanDev) <= kPlanBytes, "plan header");

// One work item of the tile pass - a list, or one slice of a shared list - as the plan kernel writes it and a wave of
// the tile pass reads it with one scalar load (decoding an item number cost five integer divisions per item).
// Item order: a GROUP is one tile of one image with all its heads and slices (M * ksplit items that read the same
// query rows); groups are numbered level block by level block (longest lists first), image by image and tile row by
// tile row, and every level block is cut into 8 contiguous ranges, one per XCD (XCD = blockIdx % 8 of the waves that
// take it): every XCD gets the same mix of long and short lists, and the same band of every level - a query row is
// named by ~2 neighbouring tiles per level and by the same place on every level, so most of its re-reads (grad_out row,
// offsets / logits row) hit the L2 that already holds it (dealt round-robin, every tile's neighbours sat on other XCDs
// and each duplicate went to the fabric: 119 MiB FETCH_SIZE per extractor call).  Within an XCD the items - its groups
// in order, m-major inside a group - are dealt to the XCD's waves with stride gridDim / 8.  Placement only affects speed.
struct ItemDesc {
    int n, ml;          // image; head | level << 16
    int tyx;            // tile row | tile column << 16
    int lst;            // list index ((n * T + tile) * M + m): its counter is counter[lst * kCtrStride]
    int ent;            // first entry slot of the list
    int jks;            // slice | slices << 16
    int slab;           // first slab of the list (slices > 1)
    int pad;
};
static_assert(sizeof(ItemDesc) == 32, "item descriptor");

// Host-side bounds (no geometry needed).  Tiles of a level: ceil(H/4) * ceil(W/8) <= H*W/4 + 1 (H >= 1, W >= 1: at
// W = 1 a tile holds 4 pixels), so T <= S/4 + L; entry slots: tiles * cap with cap <= 12 Lq / tiles + 192; slices of
// shared lists: tiles * ksplit <= tiles + 2 Lq / 512 + 1 per level.
struct Bounds {
    int64_t Tmax, ETmax, nlists_max, nslabs_max, items_per_xcd;
    int64_t off_plan, off_counts, off_entries, off_slabs, off_items, off_ga, total;
};

int make_bounds(const char *fn, int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq, int64_t P, Bounds *b) {
    if (L < 1 || L > kMaxL || P != kP) return fail(VAH_E_UNSUPPORTED, "%s: the tiled path needs 1 <= L <= %d and P == %d", fn, kMaxL, kP);
    if (Lq >= (1 << 24) || S >= ((int64_t)1 << 28) || N >= 32768 || M >= 32768 || N * M * L >= 65536 ||
        N * Lq * M * L * P * 2 >= ((int64_t)1 << 31) || N * M * (S / 4 + L) * 16 >= ((int64_t)1 << 31))
        return fail(VAH_E_UNSUPPORTED, "%s: problem too large for the tiled path", fn);
    b->Tmax = S / 4 + L;
    b->ETmax = 12 * Lq * L + 192 * b->Tmax;
    b->nlists_max = N * M * b->Tmax;
    b->nslabs_max = N * M * (L * (Lq / 128 + 1));
    const int64_t items_max = N * M * (b->Tmax + L * (Lq / 256 + 1));
    b->items_per_xcd = items_max / 8 + L * M * (Lq / 512 + 2) + 8;      // an eighth of every level block: at most one group more per block
    if (b->nlists_max >= ((int64_t)1 << 26) || N * M * b->ETmax >= ((int64_t)1 << 31) || items_max >= ((int64_t)1 << 30) ||
        b->nslabs_max >= ((int64_t)1 << 30))
        return fail(VAH_E_UNSUPPORTED, "%s: problem too large for the tiled path", fn);
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    b->off_plan = 0;
    b->off_counts = kPlanBytes;
    b->off_entries = b->off_counts + up(b->nlists_max * kCtrStride * 4);
    b->off_slabs = b->off_entries + up(N * M * b->ETmax * 4);
    b->off_items = b->off_slabs + up(b->nslabs_max * kTilePx * kD * 4);
    b->off_ga = b->off_items + up(8 * b->items_per_xcd * (int64_t)sizeof(ItemDesc));
    b->total = b->off_ga + up(N * Lq * M * L * kP * 16);          // fused core: per-sample gradient records (8 or 16 bytes)
    return VAH_OK;
}

// ---- pass 0: plan + zero-fill ------------------------------------------------------------------------------------
// Every thread derives what the zero-fill needs (tiles per level: shifts and adds); ONE thread derives the rest -
// capacities, shared lists, item order, items per XCD: a dozen 32-bit divisions - and writes the plan; all of them zero
// the list counters (and, if the levels do not tile [0, S), grad_value: gv_words 16-byte words).  The item descriptors
// are written by the binning kernel's first threads, which read the plan anyway.
struct LightPlan {
    int T, nlists;
    bool partition;
    int H[kMaxL], W[kMaxL], st[kMaxL], nt[kMaxL];
    bool ok[kMaxL];
};
__device__ __forceinline__ LightPlan light_plan(const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi, int L, int N,
                                                int64_t S, int M, int64_t Tmax) {
    LightPlan lp;
    int T = 0;
    int64_t expect = 0;
    bool partition = true;
#pragma unroll
    for (int l = 0; l < kMaxL; ++l) {
        lp.H[l] = lp.W[l] = lp.st[l] = lp.nt[l] = 0, lp.ok[l] = false;
        if (l < L) {
            const int64_t H = shapes[2 * l], W = shapes[2 * l + 1], st = lsi[l];
            bool ok = H >= 1 && W >= 1 && st >= 0 && H <= S && W <= S && st + H * W <= S && H <= 32760 && W <= 32760;
            // (levels that overlap may claim more rows than S in total: one whose tiles no longer fit the workspace bounds,
            // which assume sum(H * W) <= S, is dropped like a level outside the value rows)
            const int nt = ok ? (int)((W + kTW - 1) >> kTShX) * (int)((H + kTH - 1) >> kTShY) : 0;
            ok = ok && (int64_t)T + nt <= Tmax;
            partition = partition && ok && st == expect;
            expect = st + H * W;
            // a level that is no window of the value rows has no tiles: its samples contribute nothing (as in msda.hip)
            lp.ok[l] = ok, lp.nt[l] = ok ? nt : 0, lp.H[l] = ok ? (int)H : 0, lp.W[l] = ok ? (int)W : 0, lp.st[l] = ok ? (int)st : 0;
            T += lp.nt[l];
        }
    }
    lp.partition = partition && expect == S;
    lp.T = T;
    lp.nlists = N * M * T;
    return lp;
}

__global__ __launch_bounds__(256) void msda_plan(const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi, int L,
                                                  int N, int64_t S, int M, int Lq, Bounds bd, unsigned char *__restrict__ ws,
                                                  uint4 *__restrict__ gv, int64_t gv_words) {
    const LightPlan lp = light_plan(shapes, lsi, L, N, S, M, bd.Tmax);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PlanDev g{};
        g.L = L, g.T = lp.T, g.partition = lp.partition, g.nlists = lp.nlists;
        unsigned ET = 0, tb = 0, per_item[kMaxL] = {0, 0, 0, 0};
        int nslabs = 0;
#pragma unroll
        for (int l = 0; l < kMaxL; ++l) {
            if (l >= L) continue;
            const unsigned nt = (unsigned)lp.nt[l], lq = (unsigned)Lq;
            g.valid[l] = lp.ok[l], g.H[l] = lp.H[l], g.W[l] = lp.W[l], g.start[l] = lp.st[l];
            g.ntx[l] = lp.ok[l] ? (lp.W[l] + kTW - 1) >> kTShX : 0;
            g.ntiles[l] = (int)nt;
            g.tbase[l] = (int)tb;
            // capacity: 12 rows' worth per tile of evenly spread queries (a (q, level) row names ~2 tiles when its samples
            // sit close together - the adapter - and ~5.6 when each sample falls somewhere else - test.py's uniform recipe)
            unsigned cap = nt ? (12u * lq + nt - 1) / nt : 128u;
            cap = (cap < 128u ? 128u : cap + 63u) / 64u * 64u;
            const unsigned all = (lq + 63u) / 64u * 64u < 128u ? 128u : (lq + 63u) / 64u * 64u;
            if (cap > all) cap = all;                        // a list never holds more than every query
            g.cap[l] = (int)cap;
            g.ebase[l] = (int)ET;
            // shared lists: by the even load (2 tiles per row), not by the capacity
            unsigned even = nt ? (2u * lq + nt - 1) / nt : 0u;
            even = even < cap ? even : cap;
            unsigned ks = (even / 64u + kChunksPerWg) / kChunksPerWg;
            if ((int64_t)nslabs + (int64_t)N * nt * M * ks > bd.nslabs_max) ks = 1;       // cannot happen for ks from the even load; guard
            g.ksplit[l] = (int)ks;
            per_item[l] = even / ks;
            g.slab[l] = nslabs;
            if (ks > 1) nslabs += N * (int)nt * M * (int)ks;
            tb += nt;
            ET += nt * cap;
        }
        g.ET = (int)ET;
        // item order: the levels with the longest lists first (their items run longest: no tail of a few long lists at
        // the end of the launch)
        int order[kMaxL] = {0, 1, 2, 3};
#pragma unroll
        for (int a = 1; a < kMaxL; ++a)
#pragma unroll
            for (int b = a; b > 0; --b)
                if (a < L && per_item[order[b]] > per_item[order[b - 1]]) {
                    const int t = order[b];
                    order[b] = order[b - 1];
                    order[b - 1] = t;
                }
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) g.wgorder[i] = order[i];
        // groups / items per level block and per XCD (see ItemDesc)
        int gb = 0, nw = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) g.xcount[x] = 0;
#pragma unroll
        for (int b = 0; b < kMaxL; ++b) {
            const int l = order[b];
            const int ng = b < L ? N * g.ntiles[l] : 0, per = M * (b < L ? g.ksplit[l] : 1);
            g.gb[b] = gb, g.ib[b] = nw;
#pragma unroll
            for (int x = 0; x < 8; ++x) g.xcount[x] += ((((x + 1) * ng) >> 3) - ((x * ng) >> 3)) * per;   // eighth x of the block
            gb += ng;
            nw += ng * per;
        }
        g.gb[kMaxL] = gb, g.ib[kMaxL] = nw;
        g.nwgs = nw;
        *reinterpret_cast<PlanDev *>(ws + bd.off_plan) = g;
    }
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
    uint4 *cz = reinterpret_cast<uint4 *>(ws + bd.off_counts);
    const int64_t cwords = (int64_t)lp.nlists * kCtrStride / 4;
    for (int64_t i = tid; i < cwords; i += nthr) cz[i] = make_uint4(0, 0, 0, 0);
    if (!lp.partition)
        for (int64_t i = tid; i < gv_words; i += nthr) gv[i] = make_uint4(0, 0, 0, 0);
}

// descriptor of item i (global numbering: level blocks, groups, heads, slices) and its place in the XCD tables
__device__ __forceinline__ void write_item(const PlanDev &g, const Bounds &bd, ItemDesc *__restrict__ items, int i, int M) {
    int blk = 0;
#pragma unroll
    for (int b = 1; b < kMaxL; ++b) blk = (b < g.L && i >= g.ib[b]) ? b : blk;
    const int l = blk == 0 ? g.wgorder[0] : blk == 1 ? g.wgorder[1] : blk == 2 ? g.wgorder[2] : g.wgorder[3];
    const int gb_b = blk == 0 ? g.gb[0] : blk == 1 ? g.gb[1] : blk == 2 ? g.gb[2] : g.gb[3];
    const int ib_b = blk == 0 ? g.ib[0] : blk == 1 ? g.ib[1] : blk == 2 ? g.ib[2] : g.ib[3];
    const int ks = l == 0 ? g.ksplit[0] : l == 1 ? g.ksplit[1] : l == 2 ? g.ksplit[2] : g.ksplit[3];
    const int nt = l == 0 ? g.ntiles[0] : l == 1 ? g.ntiles[1] : l == 2 ? g.ntiles[2] : g.ntiles[3];
    const int ntx = l == 0 ? g.ntx[0] : l == 1 ? g.ntx[1] : l == 2 ? g.ntx[2] : g.ntx[3];
    const int tbase = l == 0 ? g.tbase[0] : l == 1 ? g.tbase[1] : l == 2 ? g.tbase[2] : g.tbase[3];
    const int cap = l == 0 ? g.cap[0] : l == 1 ? g.cap[1] : l == 2 ? g.cap[2] : g.cap[3];
    const int ebase = l == 0 ? g.ebase[0] : l == 1 ? g.ebase[1] : l == 2 ? g.ebase[2] : g.ebase[3];
    const int slab0 = l == 0 ? g.slab[0] : l == 1 ? g.slab[1] : l == 2 ? g.slab[2] : g.slab[3];
    unsigned r_ = (unsigned)(i - ib_b);
    const int j = (int)(r_ % (unsigned)ks);
    r_ /= (unsigned)ks;
    const int m = (int)(r_ % (unsigned)M);
    r_ /= (unsigned)M;                                  // group inside the block = n * ntiles + tile
    const int tl = (int)(r_ % (unsigned)nt), n = (int)(r_ / (unsigned)nt);
    // XCD of the group: eighth x of its level block, groups [x * ng / 8, (x + 1) * ng / 8)
    const int ng_b = (blk == 0 ? g.gb[1] : blk == 1 ? g.gb[2] : blk == 2 ? g.gb[3] : g.gb[4]) - gb_b;
    const int x = (int)((8u * (r_ + 1u) + (unsigned)ng_b - 1u) / (unsigned)ng_b) - 1;
    // local position on XCD x: its items of earlier blocks + its groups of this block before this one
    int pos = 0;
#pragma unroll
    for (int b = 0; b < kMaxL; ++b) {
        const int ng = g.gb[b + 1] - g.gb[b];
        const int lb = g.wgorder[b];
        const int per = M * (b < g.L ? (lb == 0 ? g.ksplit[0] : lb == 1 ? g.ksplit[1] : lb == 2 ? g.ksplit[2] : g.ksplit[3]) : 1);
        if (b < blk) pos += ((((x + 1) * ng) >> 3) - ((x * ng) >> 3)) * per;
        if (b == blk) pos += ((int)r_ - ((x * ng) >> 3)) * per;
    }
    pos += m * ks + j;
    ItemDesc d;
    d.n = n, d.ml = m | (l << 16);
    const int ty = (int)((unsigned)tl / (unsigned)ntx), tx = tl - ty * ntx;
    d.tyx = ty | (tx << 16);
    d.lst = (n * g.T + tbase + tl) * M + m;
    d.ent = ((n * g.ET + ebase + tl * cap) * M) + m * cap;
    d.jks = j | (ks << 16);
    d.slab = slab0 + ((n * nt + tl) * M + m) * ks;
    d.pad = 0;
    if (pos < bd.items_per_xcd) items[(int64_t)x * bd.items_per_xcd + pos] = d;
}

// ---- sample sources ---------------------------------------------------------------------------
// A source hands out one (n, q, m) row's samples of one level in two steps: load() only issues the
// global loads - whole 16 / 8-byte vectors, kept as raw words so that a caller can hold the NEXT chunk's
// operands in few registers while it works on the current one - and xy() / weights() decode them.
// Plain: sampling_locations (N,Lq,M,L,P,2) and attention_weights (N,Lq,M,L,P), fp32 (the reference API).
struct PlainSrc {
    static constexpr int kLevels = 0;       // run-time
    const float *loc, *attn;
    int LP;
    float *grad_loc, *grad_attn;           // outputs of the tile pass (spec cuh:156-158)
    // gradients of the P samples of (row, level): own = bit p set when this caller owns sample p
    __device__ __forceinline__ void store_grads(int row, int l, unsigned own, const float (&gx)[kP], const float (&gy)[kP],
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
    __device__ __forceinline__ Raw load(int row, int q, int l) const {
        Raw r;
        const float4 *lp = reinterpret_cast<const float4 *>(loc + (row * LP + l * kP) * 2);       // 32-byte aligned
        r.xy[0] = lp[0];
        r.xy[1] = lp[1];
        r.a = WEIGHTS ? *reinterpret_cast<const float4 *>(attn + row * LP + l * kP) : make_float4(0.f, 0.f, 0.f, 0.f);
        return r;
    }
    struct LevelConst {};
    __device__ __forceinline__ LevelConst level_const(int H, int W) const { return LevelConst{}; }
    __device__ __forceinline__ float2 xy(const Raw &r, int p, const LevelConst &) const {
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
// PT: type of offsets / logits; GPT: type their gradients are written in (the module's Linear layers take bf16
// gradients under autocast also when their outputs are kept in fp32).
template <typename PT, typename GPT, int L>
struct FusedSrc {
    const PT *off, *logit;
    const float *ref;
    int ref_levels;
    int os, ls;                            // elements between the offsets / logits of consecutive (n, q, m) rows
    static constexpr int kLevels = L;
    static constexpr int LP = L * kP;
    static constexpr int OW = kP * 2 * (int)sizeof(PT) / 4;      // words of one level's offsets (4 or 8)
    static constexpr int LW = LP * (int)sizeof(PT) / 4;          // words of the row's logits (even)
    // The tile pass writes one RECORD per owned sample - {d(offset x), d(offset y), d(out)/d(attention probability)} -
    // into a scratch (N,Lq,M,L*P) of 8-byte (bf16 gradients) / 16-byte (fp32 gradients) records: a tile owns some of a
    // row's samples, and every partial store instruction costs a wave 64 cache-line visits whatever its width - one
    // 8-byte store per owned sample instead of a 4-byte d(offset) and a 4-byte d(p) store.  msda_grad_finish turns
    // the records into d_offsets and (softmax backward) d_logits.
    static constexpr int RW = sizeof(GPT) == 2 ? 2 : 4;          // 32-bit words per record
    uint32_t *rec;
    __device__ __forceinline__ void pack(float gx, float gy, float ga_, uint32_t (&w)[RW]) const {
        if constexpr (RW == 2) {
            bf16x2 o;
            o[0] = (__bf16)gx, o[1] = (__bf16)gy;
            w[0] = __builtin_bit_cast(uint32_t, o), w[1] = __builtin_bit_cast(uint32_t, ga_);
        } else {
            w[0] = __builtin_bit_cast(uint32_t, gx), w[1] = __builtin_bit_cast(uint32_t, gy);
            w[2] = __builtin_bit_cast(uint32_t, ga_), w[3] = 0u;
        }
    }
    __device__ __forceinline__ void store_grads(int row, int l, unsigned own, const float (&gx)[kP], const float (&gy)[kP],
                                                const float (&gav)[kP], int H, int W) const {
        uint32_t *rp = rec + (size_t)(unsigned)(row * LP + l * kP) * RW;
        uint32_t w[kP][RW];
#pragma unroll
        for (int p = 0; p < kP; ++p) pack(gx[p], gy[p], gav[p], w[p]);
        if (own == 0xFu) {
            if constexpr (RW == 2) {
                *reinterpret_cast<uint4 *>(rp) = make_uint4(w[0][0], w[0][1], w[1][0], w[1][1]);
                *reinterpret_cast<uint4 *>(rp + 4) = make_uint4(w[2][0], w[2][1], w[3][0], w[3][1]);
            } else {
#pragma unroll
                for (int p = 0; p < kP; ++p) *reinterpret_cast<uint4 *>(rp + 4 * p) = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
            }
        } else {
#pragma unroll
            for (int p = 0; p < kP; ++p)
                if ((own >> p) & 1) {
                    if constexpr (RW == 2) *reinterpret_cast<uint2 *>(rp + 2 * p) = make_uint2(w[p][0], w[p][1]);
                    else *reinterpret_cast<uint4 *>(rp + 4 * p) = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
                }
        }
    }
    struct Raw {
        uint32_t o[OW];
        uint32_t lg[LW];
        float2 rp;
    };
    template <bool WEIGHTS>
    __device__ __forceinline__ Raw load(int row, int q, int l) const {
        Raw r;
        const uint4 *op = reinterpret_cast<const uint4 *>(off + row * os + l * kP * 2);           // 16-byte aligned
#pragma unroll
        for (int i = 0; i < OW / 4; ++i) {
            const uint4 v = op[i];
            r.o[4 * i] = v.x, r.o[4 * i + 1] = v.y, r.o[4 * i + 2] = v.z, r.o[4 * i + 3] = v.w;
        }
        r.rp = *reinterpret_cast<const float2 *>(ref + (q * ref_levels + (ref_levels > 1 ? l : 0)) * 2);
        const uint2 *lp = reinterpret_cast<const uint2 *>(logit + row * ls);                       // 8-byte aligned
#pragma unroll
        for (int i = 0; i < LW / 2; ++i) {
            const uint2 v = WEIGHTS ? lp[i] : make_uint2(0, 0);
            r.lg[2 * i] = v.x, r.lg[2 * i + 1] = v.y;
        }
        return r;
    }
    // off / (W, H) exactly as msda_fused.hip forms it: a true division - or, for a power of two (every level of a
    // square power-of-two image), the multiplication by the exact reciprocal, which is the same number
    struct LevelConst {
        float Wf, Hf, rW, rH;
        bool pow2;
    };
    __device__ __forceinline__ LevelConst level_const(int H, int W) const {
        LevelConst c;
        c.Wf = (float)W, c.Hf = (float)H;
        c.pow2 = (W & (W - 1)) == 0 && (H & (H - 1)) == 0 && W > 0 && H > 0;
        c.rW = 1.f / c.Wf, c.rH = 1.f / c.Hf;
        return c;
    }
    __device__ __forceinline__ float2 xy(const Raw &r, int p, const LevelConst &c) const {
        const float ox = word_elem<PT>(r.o, 2 * p), oy = word_elem<PT>(r.o, 2 * p + 1);
        if (c.pow2) return make_float2(r.rp.x + ox * c.rW, r.rp.y + oy * c.rH);
        return make_float2(r.rp.x + ox / c.Wf, r.rp.y + oy / c.Hf);
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
            if constexpr (L == 1) {
                v = pr[p];
            } else {
#pragma unroll
                for (int s = 0; s < LP; ++s) v = (s == l * kP + p) ? pr[s] : v;      // l is not a compile-time constant
            }
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

// list (n, tile t of level l, head m): index ((n * T + t) * M + m); its entries start at
// ((n * ET + ebase[l] + (t - tbase[l]) * cap[l]) * M + m * cap[l]
__device__ __forceinline__ int64_t list_entry_base(const PlanDev &g, int n, int l, int t_all, int M, int m) {
    return ((int64_t)n * g.ET + g.ebase[l] + (int64_t)(t_all - g.tbase[l]) * g.cap[l]) * M + (int64_t)m * g.cap[l];
}

// ---- pass 1: binning ----------------------------------------------------------------------------
// grid (ceil(Lq / 256), N * M * L): a workgroup is 256 neighbouring queries of one (n, m, level).
// Per thread: the tiles its 4 samples touch (per sample the 1 x 1 .. 2 x 2 block of tiles its corners fall in) as a
// 32-bit mask over a window of 4 (x) x 8 (y) tiles anchored at the smallest tile row / column of the row's samples:
// a few shifts per sample instead of pairwise comparisons of 16 candidate tiles (round 2: ~1000 VALU instructions per
// thread, 17-20 us per call).  A row whose samples are spread wider (> 32 pixels apart) takes the exact walk
// below.  The workgroup counts per tile in LDS, reserves each touched list's range with ONE global atomic per tile -
// all of them in flight together - and the threads then write their entries (second LDS count for the positions).
template <typename Src, bool TAPS>
__global__ __launch_bounds__(256) void msda_bin(Src src, const unsigned char *__restrict__ ws, Bounds bd, int M, int Lq) {
    __shared__ int s_cnt[kBinTable], s_base[kBinTable];
    const PlanDev &g = *reinterpret_cast<const PlanDev *>(ws + bd.off_plan);
    int *counter = reinterpret_cast<int *>(const_cast<unsigned char *>(ws) + bd.off_counts);
    int *entries = reinterpret_cast<int *>(const_cast<unsigned char *>(ws) + bd.off_entries);
    {       // the tile pass's item descriptors: one per thread, by the first g.nwgs threads of the launch
        const int64_t gt = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        const int64_t nthr = (int64_t)gridDim.x * gridDim.y * 256;
        ItemDesc *items = reinterpret_cast<ItemDesc *>(const_cast<unsigned char *>(ws) + bd.off_items);
        for (int64_t i = gt; i < g.nwgs; i += nthr) write_item(g, bd, items, (int)i, M);
    }
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < Lq;
    const int y = blockIdx.y;
    const int l = y % g.L, m = (y / g.L) % M, n = y / (g.L * M);
    const int H = g.H[l], W = g.W[l], ntx = g.ntx[l], cap = g.cap[l];
    const int nt = g.ntiles[l];
    const int row = (n * Lq + (live ? q : 0)) * M + m;
    if (nt == 0) {                                           // the level is no window of the value rows: its samples give nothing
        if (TAPS && live) {
            const float z[kP] = {0.f, 0.f, 0.f, 0.f};
            src.store_grads(row, l, 0xFu, z, z, z, 1, 1);
        }
        return;
    }
    const bool table = nt <= kBinTable;
    if (table)
        for (int i = threadIdx.x; i < nt; i += 256) s_cnt[i] = 0;
    unsigned mask = 0;
    int ay = 0, ax = 0;
    bool wide = false;
    int ya[kP], yb[kP], xa[kP], xb[kP];
    bool in[kP];
#pragma unroll
    for (int p = 0; p < kP; ++p) ya[p] = yb[p] = xa[p] = xb[p] = 0, in[p] = false;
    if (live) {
        const typename Src::Raw raw = src.template load<false>(row, q, l);
        const typename Src::LevelConst lc = src.level_const(H, W);
        unsigned orphan = 0;
        ay = 1 << 30, ax = 1 << 30;
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const float2 xy = src.xy(raw, p, lc);
            const Base b = make_base(xy.x, xy.y, H, W);
            // tile rows / columns of the corner rows y0, y0 + 1 clipped to the map (inside: -1 <= y0 <= H - 1)
            ya[p] = max(b.y0, 0) >> kTShY, yb[p] = min(b.y0 + 1, H - 1) >> kTShY;
            xa[p] = max(b.x0, 0) >> kTShX, xb[p] = min(b.x0 + 1, W - 1) >> kTShX;
            in[p] = b.inside;
            orphan |= b.inside ? 0u : 1u << p;          // no corner anywhere: no tile will own this sample
            ay = b.inside ? min(ay, ya[p]) : ay;
            ax = b.inside ? min(ax, xa[p]) : ax;
        }
        int my = 0, mx = 0;
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const int ry0 = ya[p] - ay, ry1 = yb[p] - ay, rx0 = xa[p] - ax, rx1 = xb[p] - ax;
            my = in[p] ? max(my, ry1) : my;
            mx = in[p] ? max(mx, rx1) : mx;
            const unsigned bits = (1u << ((ry0 & 7) * 4 + (rx0 & 3))) | (1u << ((ry0 & 7) * 4 + (rx1 & 3))) |
                                  (1u << ((ry1 & 7) * 4 + (rx0 & 3))) | (1u << ((ry1 & 7) * 4 + (rx1 & 3)));
            mask |= in[p] ? bits : 0u;
        }
        wide = my > 7 || mx > 3;
        // a sample inside the gate has its corner (max(y0,0), max(x0,0)) in the map, so some tile owns it and writes
        // its gradients; the others (gate failed: spec cuh:288) get their zeros here
        if (TAPS && orphan) {
            const float z[kP] = {0.f, 0.f, 0.f, 0.f};
            src.store_grads(row, l, orphan, z, z, z, H, W);
        }
    }
    const int64_t list0 = ((int64_t)n * g.T + g.tbase[l]) * M + m;         // + tile * M
    const int64_t ent0 = ((int64_t)n * g.ET + g.ebase[l]) * M + (int64_t)m * cap;      // + tile * cap * M
    // exact walk for a row spread over more than the mask's window: every tile of every sample unless an earlier sample
    // of the row already named it; straight global atomics (rare: offsets of tens of pixels)
    if (wide) {
        mask = 0;
#pragma unroll
        for (int p = 0; p < kP; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int tyc = (c >> 1) ? yb[p] : ya[p], txc = (c & 1) ? xb[p] : xa[p];
                bool fresh = in[p] && ((c >> 1) == 0 || yb[p] != ya[p]) && ((c & 1) == 0 || xb[p] != xa[p]);
#pragma unroll
                for (int e = 0; e < kP; ++e)
                    if (e < p) fresh = fresh && !(in[e] && tyc >= ya[e] && tyc <= yb[e] && txc >= xa[e] && txc <= xb[e]);
                if (fresh) {
                    const int t = tyc * ntx + txc;
                    const int pos = atomicAdd(counter + (list0 + (int64_t)t * M) * kCtrStride, 1);
                    if (pos < cap) entries[ent0 + (int64_t)t * cap * M + pos] = q;
                }
            }
    }
    if (table) {
        __syncthreads();
        for (unsigned mk = mask; mk; mk &= mk - 1) {
            const int b = __builtin_ctz(mk);
            atomicAdd(&s_cnt[(ay + (b >> 2)) * ntx + ax + (b & 3)], 1);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += 256) {
            const int c = s_cnt[i];
            if (c) {
                s_base[i] = atomicAdd(counter + (list0 + (int64_t)i * M) * kCtrStride, c);
                s_cnt[i] = 0;
            }
        }
        __syncthreads();
        for (unsigned mk = mask; mk; mk &= mk - 1) {
            const int b = __builtin_ctz(mk);
            const int t = (ay + (b >> 2)) * ntx + ax + (b & 3);
            const int pos = s_base[t] + atomicAdd(&s_cnt[t], 1);
            if (pos < cap) entries[ent0 + (int64_t)t * cap * M + pos] = q;
        }
    } else {
        for (unsigned mk = mask; mk; mk &= mk - 1) {
            const int b = __builtin_ctz(mk);
            const int t = (ay + (b >> 2)) * ntx + ax + (b & 3);
            const int pos = atomicAdd(counter + (list0 + (int64_t)t * M) * kCtrStride, 1);
            if (pos < cap) entries[ent0 + (int64_t)t * cap * M + pos] = q;
        }
    }
}

// ---- pass 2: the tile pass -------------------------------------------------------------------------------------------
// LDS per wave: R = Wt[32 px + 1 dummy][WS] fp32 (aliased, bf16 rows with TAPS, by Dd[32 entries][DP] fp32) + G[64 entries][32 ch].
//   bf16: WS = 68 (272-byte rows: the 8-float fragment reads are conflict-free ds_read_b128)
//   fp32: WS = 65 (the one-float fragment reads of v_mfma_f32_32x32x2_f32 are conflict-free)
template <typename GT>
struct TileLds {
    static constexpr int WS = std::is_same<GT, float>::value ? 65 : 68;
    static constexpr int DP = 68;                                          // floats per entry row of Dd (60 window pixels)
    static constexpr int w_floats = ((kTilePx + 1) * WS + 3) / 4 * 4;      // + one dummy row for the corners outside the tile
    static constexpr int r_floats = w_floats > 32 * DP ? w_floats : 32 * DP;
    static constexpr int g_bytes = 64 * kD * (int)sizeof(GT);
    static constexpr int bytes = (r_floats * 4 + g_bytes + 15) / 16 * 16;
};

// phase boundary inside a single-wave workgroup: LDS hand-off between lanes + a stop for the instruction scheduler
// (without it the compiler interleaves the phases of a chunk and keeps all their operands live at once: 220 registers)
#define VAH_PHASE()                              \
    do {                                         \
        __builtin_amdgcn_wave_barrier();         \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

// one work item, as the tile pass uses it (wave-uniform)
struct Item {
    int l, m, ks, j, H, W, ty, tx, start, cap, n, lst, slab;
    const int *list;
};

__device__ __forceinline__ int pick(const int (&a)[kMaxL], int l) { return l == 0 ? a[0] : l == 1 ? a[1] : l == 2 ? a[2] : a[3]; }

__device__ __forceinline__ Item load_item(const PlanDev &g, const ItemDesc *__restrict__ tab, const int *entries, int s) {
    const ItemDesc d = tab[s];
    Item it;
    it.n = d.n, it.m = d.ml & 0xFFFF, it.l = d.ml >> 16;
    it.ty = d.tyx & 0xFFFF, it.tx = d.tyx >> 16;
    it.lst = d.lst;
    it.list = entries + d.ent;
    it.j = d.jks & 0xFFFF, it.ks = d.jks >> 16;
    it.slab = d.slab;
    it.H = pick(g.H, it.l), it.W = pick(g.W, it.l), it.start = pick(g.start, it.l), it.cap = pick(g.cap, it.l);
    return it;
}

// what an item has to do, once its list's counter is known (wave-uniform)
struct Run {
    int nent, nchunks, nslices;
    bool scan_all, active;
};
__device__ __forceinline__ Run make_run(const Item &it, int count, int Lq, bool valid) {
    Run rn;
    rn.scan_all = count > it.cap;                       // the list overflowed: walk every query of this (n, head, level)
    rn.nent = rn.scan_all ? Lq : count;
    rn.nchunks = (rn.nent + 63) / 64;
    // slice j of the list's ks items takes chunks j, j + ks, ...; only min(ks, nchunks) of them have any
    // (at least one - slice 0 - so that an empty tile is still stored)
    rn.nslices = min(it.ks, max(rn.nchunks, 1));
    rn.active = valid && it.j < rn.nslices;
    return rn;
}

// TAPS: the tile also computes d(location) / d(attention) of the samples it owns (bf16 rows only; else a gather kernel
// of msda.hip / msda_fused.hip does).  WPS: waves per SIMD the register budget is set for.
template <typename GT, typename OT, typename Src, bool TAPS, int WPS, bool EARLY>
__global__ __launch_bounds__(64, WPS) void msda_tile(Src src, const PlanDev *__restrict__ plan, int *__restrict__ counter,
                                                     const int *__restrict__ entries, float *__restrict__ slabs,
                                                     const ItemDesc *__restrict__ items, int items_per_xcd, int M, int Lq, int64_t S,
                                                     const GT *__restrict__ value, const GT *__restrict__ grad_out,
                                                     OT *__restrict__ grad_value) {
    using LD = TileLds<GT>;
    constexpr int WS = LD::WS, DP = LD::DP;
    constexpr bool F32 = std::is_same<GT, float>::value;
    static_assert(!(TAPS && F32), "in-tile dot products are built for bf16 rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    float *Wt = reinterpret_cast<float *>(smem);
    float *Dd = reinterpret_cast<float *>(smem);                 // alias: used between the chunk's start and its zero-fill
    GT *Gs = reinterpret_cast<GT *>(smem + (size_t)LD::r_floats * 4);
    const PlanDev &g = *plan;

    // items of this wave: XCD x (observed: blockIdx % 8) has its own item table, dealt to its waves with stride gridDim / 8
    const int xcd = (int)(blockIdx.x % 8);
    const int total = xcd == 0 ? g.xcount[0] : xcd == 1 ? g.xcount[1] : xcd == 2 ? g.xcount[2] : xcd == 3 ? g.xcount[3]
                    : xcd == 4 ? g.xcount[4] : xcd == 5 ? g.xcount[5] : xcd == 6 ? g.xcount[6] : g.xcount[7];
    const ItemDesc *tab = items + (int64_t)xcd * items_per_xcd;
    const int gx = gridDim.x / 8;
    int slot = blockIdx.x / 8;
    if (slot >= total) return;
    constexpr int NV = kD * (int)sizeof(GT) / 16;               // 16-byte pieces of a grad_out row
    const int r = lane & 31, h = lane >> 5;

    // Every load of the chunk loop is UNCONDITIONAL and the loop body has no branch around a load: the compiler then
    // counts its waits (a load behind a condition is waited on with vmcnt(0), which drains the prefetches with it).
    // ---- header of an item: its list's counter and the entries of the slice's first two chunks (every list has >= 128
    // slots: the addresses are clamped into the list, the values past the count are garbage and never used), requested
    // TWO items ahead
    struct Head {
        int cnt, ea, eb;
    };
    auto load_head = [&](const Item &it) __attribute__((always_inline)) -> Head {
        Head hd;
        hd.cnt = counter[(int64_t)it.lst * kCtrStride];
        hd.ea = it.list[min(it.j * 64 + lane, it.cap - 1)];
        hd.eb = it.list[min((it.j + it.ks) * 64 + lane, it.cap - 1)];
        return hd;
    };
    // entries of the slice's first two chunks and the live mask of the first, from the header
    auto first_entries = [&](const Item &it, const Run &rn, const Head &hd, int &ea, int &eb, bool &lv) __attribute__((always_inline)) {
        const int ia = it.j * 64 + lane, ib = (it.j + it.ks) * 64 + lane;
        const int last_i = max(rn.nent - 1, 0);
        ea = rn.scan_all ? min(ia, last_i) : (ia < rn.nent ? hd.ea : 0);
        eb = rn.scan_all ? min(ib, last_i) : (ib < rn.nent ? hd.eb : 0);
        lv = rn.active && ia < rn.nent;
    };
    // value fragments of an item's window for the dot products: lane (px = r + 32 pb, half h) holds value[px][16 s + 8 h ..]
    struct VFrag {
        bf16x8 v00, v01, v10, v11;      // [pixel block][channel half]
    };
    auto load_vfrag = [&](int n_, int m_, int start_, int ty_, int tx_, int H_, int W_) __attribute__((always_inline)) -> VFrag {
        VFrag f{};
        if constexpr (TAPS) {
            const GT *vmap = value + ((int64_t)n_ * S + start_) * ((int64_t)M * kD) + m_ * kD;
            uint4 v[2][2];
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int px = r + 32 * pb;
                const int wy = px / kWinW, wx = px - wy * kWinW;
                const int yy = ty_ * kTH - 1 + wy, xx = tx_ * kTW - 1 + wx;
                const bool in = px < kWinPx && yy >= 0 && yy < H_ && xx >= 0 && xx < W_;
                const unsigned voff = (unsigned)(min(max(yy, 0), H_ - 1) * W_ + min(max(xx, 0), W_ - 1)) * (unsigned)(M * kD * (int)sizeof(GT));
                const unsigned char *vrow = reinterpret_cast<const unsigned char *>(vmap) + voff;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(vrow + 32 * s + 16 * h);
                    v[pb][s] = in ? t : make_uint4(0, 0, 0, 0);
                }
            }
            f.v00 = __builtin_bit_cast(bf16x8, v[0][0]), f.v01 = __builtin_bit_cast(bf16x8, v[0][1]);
            f.v10 = __builtin_bit_cast(bf16x8, v[1][0]), f.v11 = __builtin_bit_cast(bf16x8, v[1][1]);
        }
        return f;
    };
    VFrag vf{};
    // operands of one chunk, as loaded.  The requests for the NEXT chunk (same item, or the next item's first) go out in
    // the middle of the current chunk - after its dot products, which need the most registers - into the registers the
    // current chunk's decode / staging have freed
    int row = 0, e_n = 0;
    typename Src::Raw raw{};
    // (a clang vector, not a C array: as an array the compiler kept it - and the raw words - in scratch memory)
    typedef uint32_t gvec_t __attribute__((ext_vector_type(4 * NV)));
    gvec_t grv = {};
    bool live = false;
    // grad_out rows: COOPERATIVE loads - lane NV * i + k takes piece k of the row of the entry that lane (64 / NV) * j + i
    // holds (its index comes over by ds_bpermute): one load instruction then visits 64 / NV cache lines instead of 64
    // (the address unit takes a cycle per line whatever the width: the four 16-byte pieces of a lane's own row cost
    // 4 x 64 line visits per chunk, 40 % of the kernel's memory pipeline time).  The rows are only needed in LDS.
    constexpr int RPI = 64 / NV;                                // rows per load instruction
    auto request = [&](int n_, int m_, int l_, int e) __attribute__((always_inline)) {
        row = (n_ * Lq + e) * M + m_;
        raw = src.template load<true>(row, e, l_);
#if VAH_GR_COOP
        const unsigned char *gbytes = reinterpret_cast<const unsigned char *>(grad_out) + (size_t)(lane % NV) * 16;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e_src = __shfl(e, RPI * i + lane / NV, 64);
            const unsigned row_src = (unsigned)((n_ * Lq + e_src) * M + m_);
            const uint4 t_ = *reinterpret_cast<const uint4 *>(gbytes + (size_t)row_src * (kD * sizeof(GT)));
            grv[4 * i] = t_.x, grv[4 * i + 1] = t_.y, grv[4 * i + 2] = t_.z, grv[4 * i + 3] = t_.w;
        }
#else
        const uint4 *src4 = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(grad_out) + (size_t)(unsigned)row * (kD * sizeof(GT)));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint4 t_ = src4[i];
            grv[4 * i] = t_.x, grv[4 * i + 1] = t_.y, grv[4 * i + 2] = t_.z, grv[4 * i + 3] = t_.w;
        }
#endif
    };

    if constexpr (!TAPS) {          // Wt starts as zeros; every chunk puts back what it wrote
        for (int i = lane * 4; i < LD::w_floats; i += 64 * 4) *reinterpret_cast<float4 *>(Wt + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    Item cur = load_item(g, tab, entries, slot);
    Head hd = load_head(cur);
    int nxt_s = slot + gx;
    Head hd_n = load_head(load_item(g, tab, entries, nxt_s < total ? nxt_s : slot));
    Run run = make_run(cur, __builtin_amdgcn_readfirstlane(hd.cnt), Lq, true);
    {
        int ea, eb;
        first_entries(cur, run, hd, ea, eb, live);
        e_n = eb;
        request(cur.n, cur.m, cur.l, ea);
        vf = load_vfrag(cur.n, cur.m, cur.start, cur.ty, cur.tx, cur.H, cur.W);
    }

    while (true) {
        // ---- the header of the item after next goes out now; the next item's has had a whole item to arrive
        const bool has_n = nxt_s < total;
        const int nn_s = nxt_s + gx;
        const Item nx = load_item(g, tab, entries, has_n ? nxt_s : slot);
        const Head hd_nn = load_head(load_item(g, tab, entries, nn_s < total ? nn_s : (has_n ? nxt_s : slot)));
        const Run run_n = make_run(nx, __builtin_amdgcn_readfirstlane(hd_n.cnt), Lq, has_n);
        int ea_n, eb_n;
        bool live_first_n;
        first_entries(nx, run_n, hd_n, ea_n, eb_n, live_first_n);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int nloops = (run.active && !(ablate & 64)) ? run.nchunks : 0;
        if (cur.j >= nloops) {          // nothing to do here (empty list, idle slice): the next item's first operands go out now
            live = live_first_n, e_n = eb_n;
            request(nx.n, nx.m, nx.l, ea_n);
#if VAH_VF_PER_CHUNK
            vf = load_vfrag(nx.n, nx.m, nx.start, nx.ty, nx.tx, nx.H, nx.W);
#endif
        } else {
            const typename Src::LevelConst lc = src.level_const(cur.H, cur.W);
            const int STEP = cur.ks;
            const int last_i = max(run.nent - 1, 0);
            for (int ch = cur.j; ch < nloops; ch += STEP) {
                const bool last = ch + STEP >= nloops;
                // ---- the entries' grad_out rows -> Gs[entry][0..31] (piece lane % NV of row RPI * i + lane / NV, as loaded;
                //      lanes past the list's end hold a copy of a real row: their weights are zero, their sums never stored)
                {
#if VAH_GR_COOP
                    unsigned char *gs = reinterpret_cast<unsigned char *>(Gs) + (lane / NV) * (kD * (int)sizeof(GT)) + (lane % NV) * 16;
#pragma unroll
                    for (int i = 0; i < NV; ++i)
                        *reinterpret_cast<uint4 *>(gs + RPI * i * (kD * (int)sizeof(GT))) = make_uint4(grv[4 * i], grv[4 * i + 1], grv[4 * i + 2], grv[4 * i + 3]);
#else
                    uint4 *dst4 = reinterpret_cast<uint4 *>(Gs + lane * kD);
#pragma unroll
                    for (int i = 0; i < NV; ++i) dst4[i] = make_uint4(grv[4 * i], grv[4 * i + 1], grv[4 * i + 2], grv[4 * i + 3]);
#endif
                }
                // ---- the entry's samples relative to the tile
                float a[kP];
                src.weights(raw, cur.l, a);
                int ry[kP], rx[kP];
                float lh[kP], lw[kP];
                bool on[kP];
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    const float2 xy = src.xy(raw, p, lc);
                    const Base b = make_base(xy.x, xy.y, cur.H, cur.W);
                    ry[p] = b.y0 - cur.ty * kTH, rx[p] = b.x0 - cur.tx * kTW;
                    lh[p] = b.lh, lw[p] = b.lw;
                    on[p] = live && b.inside;
                }
                const int row_c = row;
                auto issue_next = [&]() __attribute__((always_inline)) {
                    const int idx2 = min((ch + 2 * STEP) * 64 + lane, last_i);
                    const int e_load = cur.list[min(idx2, cur.cap - 1)];
                    const int e_req = last ? ea_n : e_n;
                    live = last ? live_first_n : ((ch + STEP) * 64 + lane < run.nent);
                    e_n = last ? eb_n : (run.scan_all ? idx2 : e_load);
                    request(last ? nx.n : cur.n, last ? nx.m : cur.m, last ? nx.l : cur.l, e_req);
#if VAH_VF_PER_CHUNK
                    vf = load_vfrag(last ? nx.n : cur.n, last ? nx.m : cur.m, last ? nx.start : cur.start, last ? nx.ty : cur.ty,
                                    last ? nx.tx : cur.tx, last ? nx.H : cur.H, last ? nx.W : cur.W);
#endif
                };
                // ---- requests for the chunk to come - this item's next, or the next item's first: operands and the entry
                //      after that, all unconditional - either right behind the decode (EARLY: a whole chunk to land,
                //      24 more registers live through the dot products) or behind the dot products
                if constexpr (EARLY) issue_next();
                VAH_PHASE();
                // ---- owned samples: d(out)/d(location), d(out)/d(attention) (spec cuh:126-158) from the corner dot products
                //      <grad_out row, value row>, all 64 entries x 60 window pixels at once on the matrix cores:
                //      Dd^T[px][e] = V[px][ch] x G^T[ch][e]; the accumulator has the entry on the lane and 4 consecutive
                //      window pixels per register group: 16-byte LDS stores into Dd[e][px], read back by the entry's lane
                if constexpr (TAPS) {
                    if (!(ablate & 4)) {
                    float gx_[kP], gy_[kP], gav[kP];
#pragma unroll
                    for (int eb = 0; eb < 2; ++eb) {
                        const uint4 *grow = reinterpret_cast<const uint4 *>(Gs + (32 * eb + r) * kD);
                        const bf16x8 b0 = __builtin_bit_cast(bf16x8, grow[h]), b1 = __builtin_bit_cast(bf16x8, grow[2 + h]);
                        float *drow = Dd + r * DP + 4 * h;
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb) {
                            f32x16 dd;
#pragma unroll
                            for (int i = 0; i < 16; ++i) dd[i] = 0.f;
                            dd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb ? vf.v10 : vf.v00, b0, dd, 0, 0, 0);
                            dd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb ? vf.v11 : vf.v01, b1, dd, 0, 0, 0);
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq)
                                *reinterpret_cast<float4 *>(drow + 32 * pb + 8 * gq) = make_float4(dd[4 * gq], dd[4 * gq + 1], dd[4 * gq + 2], dd[4 * gq + 3]);
                        }
                        VAH_PHASE();
                        const float *mine = Dd + r * DP;
#pragma unroll
                        for (int p = 0; p < kP; ++p) {
                            // window cell of corner 00: (ry + 1, rx + 1); owned samples have -1 <= ry <= 3, -1 <= rx <= 7
                            const int cell = min(max(ry[p] + 1, 0), kWinH - 2) * kWinW + min(max(rx[p] + 1, 0), kWinW - 2);
                            const float t0 = mine[cell], t1 = mine[cell + 1], t2 = mine[cell + kWinW], t3 = mine[cell + kWinW + 1];
                            // corner rows / columns inside the map
                            const int y0 = ry[p] + cur.ty * kTH, x0 = rx[p] + cur.tx * kTW;
                            const bool vy0 = y0 >= 0, vy1 = y0 + 1 <= cur.H - 1, vx0 = x0 >= 0, vx1 = x0 + 1 <= cur.W - 1;
                            const float d00 = (vy0 && vx0) ? t0 : 0.f, d01 = (vy0 && vx1) ? t1 : 0.f;
                            const float d10 = (vy1 && vx0) ? t2 : 0.f, d11 = (vy1 && vx1) ? t3 : 0.f;
                            const float hh = 1.f - lh[p], hw = 1.f - lw[p];
                            const float gav_ = hh * (hw * d00 + lw[p] * d01) + lh[p] * (hw * d10 + lw[p] * d11);
                            const float gy__ = (hw * (d10 - d00) + lw[p] * (d11 - d01)) * a[p];
                            const float gx__ = (hh * (d01 - d00) + lh[p] * (d11 - d10)) * a[p];
                            if (eb == 0) gav[p] = gav_, gy_[p] = gy__, gx_[p] = gx__;
                            else gav[p] = h ? gav_ : gav[p], gy_[p] = h ? gy__ : gy_[p], gx_[p] = h ? gx__ : gx_[p];
                        }
                        VAH_PHASE();
                    }
                    unsigned own = 0;
#pragma unroll
                    for (int p = 0; p < kP; ++p) {
                        // owner: the tile of the first in-map corner in the order 00 01 10 11
                        const int y0 = ry[p] + cur.ty * kTH, x0 = rx[p] + cur.tx * kTW;
                        const int fy = y0 >= 0 ? ry[p] : ry[p] + 1, fx = x0 >= 0 ? rx[p] : rx[p] + 1;
                        const bool owned = on[p] && (unsigned)fy < (unsigned)kTH && (unsigned)fx < (unsigned)kTW;
                        own |= owned ? 1u << p : 0u;
                    }
                    if (!(ablate & 16) && own) src.store_grads(row_c, cur.l, own, gx_, gy_, gav, cur.H, cur.W);
                    }
                }
                if constexpr (!EARLY) issue_next();
                if constexpr (TAPS) {
                    // ---- the region held Dd: zero it for this chunk's Wt (16-byte stores)
                    for (int i = lane * 4; i < LD::w_floats; i += 64 * 4)
                        *reinterpret_cast<float4 *>(Wt + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                    VAH_PHASE();
                }
                // ---- column `lane` of Wt: attention x bilinear weight of every corner that lands in this tile.  The four
                //      corners of a sample are four different pixels: their read-add-writes go out together; a corner
                //      outside the tile goes to the dummy row with weight 0, so nothing here branches
                float *wcell[kP][4];
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    const bool iy0 = (unsigned)ry[p] < (unsigned)kTH, iy1 = (unsigned)(ry[p] + 1) < (unsigned)kTH;
                    const bool ix0 = (unsigned)rx[p] < (unsigned)kTW, ix1 = (unsigned)(rx[p] + 1) < (unsigned)kTW;
                    const float fy0 = (on[p] && iy0) ? a[p] * (1.f - lh[p]) : 0.f, fy1 = (on[p] && iy1) ? a[p] * lh[p] : 0.f;
                    const float fx0 = ix0 ? 1.f - lw[p] : 0.f, fx1 = ix1 ? lw[p] : 0.f;
                    const int base = ry[p] * kTW + rx[p];
                    const int px00 = (iy0 && ix0) ? base : kTilePx, px01 = (iy0 && ix1) ? base + 1 : kTilePx;
                    const int px10 = (iy1 && ix0) ? base + kTW : kTilePx, px11 = (iy1 && ix1) ? base + kTW + 1 : kTilePx;
                    wcell[p][0] = Wt + px00 * WS + lane, wcell[p][1] = Wt + px01 * WS + lane;
                    wcell[p][2] = Wt + px10 * WS + lane, wcell[p][3] = Wt + px11 * WS + lane;
                    if (!(ablate & 1)) {
                        const float o0 = *wcell[p][0], o1 = *wcell[p][1], o2 = *wcell[p][2], o3 = *wcell[p][3];
                        *wcell[p][0] = o0 + fy0 * fx0;
                        *wcell[p][1] = o1 + fy0 * fx1;
                        *wcell[p][2] = o2 + fy1 * fx0;
                        *wcell[p][3] = o3 + fy1 * fx1;
                    }
                }
                VAH_PHASE();
                // ---- dV^T[32 ch, 32 px] += G^T[32 ch, 64 k] x Wt^T[64 k, 32 px]
                if (ablate & 2) {
                } else if constexpr (F32) {
#pragma unroll 8
                    for (int kk = 0; kk < 32; ++kk) {
                        const int k = 2 * kk + h;
                        const float afr = reinterpret_cast<const float *>(Gs)[k * kD + r];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr, Wt[r * WS + k], acc, 0, 0, 0);
                    }
                } else {
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
                        const s16x8 t01 = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const bf16x8 gfr = __builtin_bit_cast(bf16x8, t01);
                        const float *wp = Wt + r * WS + 16 * s + 8 * h;
                        const float4 w0 = *reinterpret_cast<const float4 *>(wp);
                        const float4 w1 = *reinterpret_cast<const float4 *>(wp + 4);
                        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                        bf16x8 whi, wlo;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            whi[jj] = (__bf16)wv[jj];
                            wlo[jj] = (__bf16)(wv[jj] - (float)whi[jj]);
                        }
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, whi, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, wlo, acc, 0, 0, 0);
                    }
                }
                VAH_PHASE();
                if constexpr (!TAPS) {
                    // ---- put the column back to zero (what this lane wrote; the dummy row may hold anything)
                    if (!(ablate & 1))
#pragma unroll
                        for (int p = 0; p < kP; ++p)
#pragma unroll
                            for (int c = 0; c < 4; ++c) *wcell[p][c] = 0.f;
                    VAH_PHASE();
                }
            }
        }

        // ---- the tile: the accumulator has the pixel on the lane (column) and the channels 8g + 4h .. + 3 in register
        // group g: stored straight from registers.  A list shared by several items goes through per-slice slabs in the
        // workspace: every slice stores its partial tile, the slice that arrives last adds them up in slice order - so
        // the sum does not depend on who arrives when - and stores the tile.
        if (run.active) {
            bool store = true;
            if (run.nslices > 1) {
                // Hand-off without cache-wide fences: every slab store is a write-through (`sc1`: relaxed agent-scope atomic
                // store), the wave drains them (s_waitcnt vmcnt(0)) before ONE lane adds to the list's arrival counter, and
                // the wave whose add comes last reads the slabs back with `sc1` loads (relaxed agent-scope atomic loads:
                // served by L2, never by a stale L1) - the measured-valid form of MI355X_MICROARCH.md "Valid forms", first
                // row, for single-wave workgroups.
                float *mine = slabs + (int64_t)cur.slab * (kTilePx * kD);
                float *dst = mine + (int64_t)cur.j * kTilePx * kD + r * kD + 4 * h;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    __hip_atomic_store(dst + 8 * (i >> 2) + (i & 3), acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int arrived = 0;
                if (lane == 0) arrived = __hip_atomic_fetch_add(counter + (int64_t)cur.lst * kCtrStride + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                arrived = __builtin_amdgcn_readfirstlane(arrived);
                store = arrived == run.nslices - 1;
                if (store) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                    for (int sl = 0; sl < run.nslices; ++sl) {
                        const float *part = mine + (int64_t)sl * kTilePx * kD + r * kD + 4 * h;
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            acc[i] += __hip_atomic_load(part + 8 * (i >> 2) + (i & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            const int yy = cur.ty * kTH + (r >> kTShX), xx = cur.tx * kTW + (r & (kTW - 1));
            if (store && yy < cur.H && xx < cur.W) {
                const int64_t stride = (int64_t)M * kD;
                OT *dst = grad_value + (((int64_t)cur.n * S + cur.start) + (int64_t)yy * cur.W + xx) * stride + cur.m * kD + 4 * h;
                if (g.partition) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        if constexpr (std::is_same<OT, float>::value) {
                            *reinterpret_cast<float4 *>(dst + 8 * gq) = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
                        } else {
                            bf16x4 o;
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] = (__bf16)acc[4 * gq + c];
                            *reinterpret_cast<bf16x4 *>(dst + 8 * gq) = o;
                        }
                    }
                } else {
                    // levels with gaps / overlaps: grad_value was zero-filled by the plan kernel, contributions are added
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if constexpr (std::is_same<OT, float>::value) {
                                atomicAdd(dst + 8 * gq + c, acc[4 * gq + c]);
                            } else if ((c & 1) == 0) {
                                bf16x2 pr;
                                pr[0] = (__bf16)acc[4 * gq + c], pr[1] = (__bf16)acc[4 * gq + c + 1];
                                __builtin_amdgcn_global_atomic_fadd_v2bf16(
                                    (__attribute__((__vector_size__(2 * sizeof(short)))) short __attribute__((address_space(1))) *)(dst + 8 * gq + c),
                                    __builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(short)))) short, pr));
                            }
                        }
                }
            }
        }
        if (!has_n) break;
        slot = nxt_s, nxt_s = nn_s;
        cur = nx, hd = hd_n, run = run_n;
        hd_n = hd_nn;
#if !VAH_VF_PER_CHUNK
        // the item's value fragments: once per item, behind the first chunk's staging and decode (60 rows of 64 bytes:
        // reloading them with every chunk's requests was a fifth of the kernel's L2 requests)
        if (run.active && run.nchunks > cur.j) vf = load_vfrag(cur.n, cur.m, cur.start, cur.ty, cur.tx, cur.H, cur.W);
#endif
    }
}

// ---- fused core only: records -> d_offsets, d_logits ------------------------------------------------------------
// d_logit[s] = p_s * (ga_s - sum_t p_t ga_t), p = softmax(logits of the (n, q, m) row), ga = d(out)/d(p) as the
// tile pass (and, for gated samples, the binning pass) left it in the records; d_offset = the records' first two
// numbers.  One thread per row.
template <typename PT, typename GPT, int L>
__global__ __launch_bounds__(256) void msda_grad_finish(const PT *__restrict__ logit, int64_t ls, const uint32_t *__restrict__ rec,
                                                        int64_t rows, GPT *__restrict__ d_off, GPT *__restrict__ d_logit, int64_t dos,
                                                        int64_t dls) {
    constexpr int LP = L * kP;
    constexpr int RW = sizeof(GPT) == 2 ? 2 : 4;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float p[LP], g[LP];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < LP; ++s) {
        p[s] = (float)logit[row * ls + s];
        mx = fmaxf(mx, p[s]);
    }
    const uint4 *rp = reinterpret_cast<const uint4 *>(rec + row * LP * RW);
    if constexpr (RW == 2) {
        uint32_t *dp = reinterpret_cast<uint32_t *>(d_off + row * dos);
#pragma unroll
        for (int s = 0; s < LP; s += 4) {
            const uint4 v0 = rp[s / 2], v1 = rp[s / 2 + 1];
            g[s] = __builtin_bit_cast(float, v0.y), g[s + 1] = __builtin_bit_cast(float, v0.w);
            g[s + 2] = __builtin_bit_cast(float, v1.y), g[s + 3] = __builtin_bit_cast(float, v1.w);
            // 8-byte stores: the rows of the module's interleaved gradient matrix are 8-byte, not 16-byte, aligned
            *reinterpret_cast<uint2 *>(dp + s) = make_uint2(v0.x, v0.z);
            *reinterpret_cast<uint2 *>(dp + s + 2) = make_uint2(v1.x, v1.z);
        }
    } else {
        float *dp = reinterpret_cast<float *>(d_off + row * dos);
#pragma unroll
        for (int s = 0; s < LP; s += 2) {
            const uint4 v0 = rp[s], v1 = rp[s + 1];
            g[s] = __builtin_bit_cast(float, v0.z), g[s + 1] = __builtin_bit_cast(float, v1.z);
            *reinterpret_cast<uint4 *>(dp + 2 * s) = make_uint4(v0.x, v0.y, v1.x, v1.y);
        }
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
    for (int s = 0; s < LP; ++s) d_logit[row * dls + s] = (GPT)(p[s] * (g[s] - dot));
}

// ---- host side -----------------------------------------------------------------------------------
// waves of the tile pass the chip holds at once (single-wave workgroups; LDS- or register-bound), per kernel
template <typename K>
int resident_waves(K kernel, int smem) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, smem) != hipSuccess || per_cu < 1) per_cu = 8;
    if (per_cu > 16) per_cu = 16;
    return per_cu * kCUs;
}

template <typename GT, typename OT, typename Src, bool TAPS>
int run_tiled(const char *fn, const Src &src, const Bounds &bd, const int64_t *shapes, const int64_t *lsi, int64_t N, int64_t M,
              int64_t L, int64_t Lq, int64_t S, const GT *value, const GT *grad_out, OT *grad_value, void *ws, hipStream_t st) {
    unsigned char *base = (unsigned char *)ws;
    const int64_t gv_words = N * S * M * kD * (int64_t)sizeof(OT) / 16;
    // plan + zero-fill: sized for the counters of the bound and, in case the levels do not tile [0, S), grad_value
    int64_t zwords = bd.nlists_max * kCtrStride / 4;
    if (gv_words > zwords) zwords = gv_words;
    int64_t pgrid = (zwords + 256 * 8 - 1) / (256 * 8);
    if (pgrid > 4096) pgrid = 4096;
    if (pgrid < 1) pgrid = 1;
    hipLaunchKernelGGL(msda_plan, dim3((unsigned)pgrid), dim3(256), 0, st, shapes, lsi, (int)L, (int)N, S, (int)M, (int)Lq, bd, base,
                       (uint4 *)grad_value, gv_words);
    if (int rc = check_launch(fn)) return rc;
    const dim3 bgrid((unsigned)((Lq + 255) / 256), (unsigned)(N * M * L));
    hipLaunchKernelGGL((msda_bin<Src, TAPS>), bgrid, dim3(256), 0, st, src, (const unsigned char *)base, bd, (int)M, (int)Lq);
    if (int rc = check_launch(fn)) return rc;
    constexpr int smem = TileLds<GT>::bytes;
    // Measured on BASELINE configs[2] (bf16): one level of long lists (extractor, 6 chunks per item) - three waves per SIMD
    // with the next chunk's requests behind the dot products (68 us; 73 with two waves and the requests a whole chunk
    // ahead: 3072 items on 2048 waves); several levels of short lists (injector, 1.3 chunks per item) - the requests right
    // behind the decode, which takes 24 more registers: two waves per SIMD (80 us against 86)
    constexpr bool EARLY = TAPS && Src::kLevels > 1;
    constexpr int WPS = EARLY ? 2 : 3;
    if (int rc = allow_dynamic_lds((const void *)msda_tile<GT, OT, Src, TAPS, WPS, EARLY>, smem, fn)) return rc;
    static const int waves = resident_waves(msda_tile<GT, OT, Src, TAPS, WPS, EARLY>, smem);
    int64_t grid = waves;
    const int64_t most = N * M * bd.Tmax * kChunksPerWg;           // never more waves than the bound of items
    if (grid > most) grid = (most + 7) / 8 * 8;
    hipLaunchKernelGGL((msda_tile<GT, OT, Src, TAPS, WPS, EARLY>), dim3((unsigned)grid), dim3(64), smem, st, src, (const PlanDev *)(base + bd.off_plan),
                       (int *)(base + bd.off_counts), (const int *)(base + bd.off_entries), (float *)(base + bd.off_slabs), (const ItemDesc *)(base + bd.off_items),
                       (int)bd.items_per_xcd, (int)M, (int)Lq,
                       S, value, grad_out, grad_value);
    return check_launch(fn);
}

// Who computes d(offsets) / d(logits)?  bf16 rows: the tile pass (matrix-core dot products against the tile's value
// window); fp32 rows: the gather kernels of msda.hip / msda_fused.hip in front (exact fp32 dot products).
template <typename VT>
constexpr bool kTapsInTile = std::is_same<VT, __bf16>::value;

template <typename VT, typename PT, typename GPT, int L>
int fused_tiled(const char *fn, const Bounds &bd, const void *value, const int64_t *shapes, const int64_t *lsi, const void *off,
                const void *logit, int64_t os, int64_t ls, const float *ref, int ref_levels, int64_t N, int64_t M, int64_t Lq, int64_t S,
                const void *grad_out, void *grad_value, int gv_bf16, void *d_off, void *d_logit, int64_t dos, int64_t dls, void *ws,
                hipStream_t st) {
    constexpr bool TAPS = kTapsInTile<VT>;
    uint32_t *rec = (uint32_t *)((char *)ws + bd.off_ga);
    FusedSrc<PT, GPT, L> src{(const PT *)off, (const PT *)logit, ref, ref_levels, (int)os, (int)ls, rec};
    int rc;
    if (gv_bf16) {
        if constexpr (std::is_same<VT, __bf16>::value)
            rc = run_tiled<VT, __bf16, FusedSrc<PT, GPT, L>, TAPS>(fn, src, bd, shapes, lsi, N, M, L, Lq, S, (const VT *)value,
                                                                 (const VT *)grad_out, (__bf16 *)grad_value, ws, st);
        else
            return fail(VAH_E_UNSUPPORTED, "%s: a bf16 grad_value needs bf16 values", fn);
    } else {
        rc = run_tiled<VT, float, FusedSrc<PT, GPT, L>, TAPS>(fn, src, bd, shapes, lsi, N, M, L, Lq, S, (const VT *)value,
                                                            (const VT *)grad_out, (float *)grad_value, ws, st);
    }
    if (rc || !TAPS) return rc;
    const int64_t rows = N * Lq * M;
    hipLaunchKernelGGL((msda_grad_finish<PT, GPT, L>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (const PT *)logit, ls,
                       (const uint32_t *)rec, rows, (GPT *)d_off, (GPT *)d_logit, dos, dls);
    return check_launch(fn);
}

}  // namespace
}  // namespace vah

extern "C" {

int64_t vah_msda_tile_ws_bytes(int64_t N, int64_t S, int64_t M, int64_t L, int64_t Lq, int64_t P) {
    vah::clear_error();
    vah::Bounds bd;
    if (vah::make_bounds("vah_msda_tile_ws_bytes", N, S, M, L, Lq, P, &bd)) return -1;
    return bd.total;
}

int vah_msda_backward_tiled_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                                const float *attn, const float *grad_out, int64_t N, int64_t S, int64_t M, int64_t D,
                                int64_t L, int64_t Lq, int64_t P, float *grad_value, float *grad_loc, float *grad_attn,
                                void *ws, int64_t ws_bytes, void *stream) {
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
    Bounds bd;
    if (int rc = make_bounds(fn, N, S, M, L, Lq, P, &bd)) return rc;
    if (ws_bytes < bd.total) return fail(VAH_E_SHAPE, "%s: workspace too small (%lld < %lld)", fn, (long long)ws_bytes, (long long)bd.total);
    hipStream_t st = (hipStream_t)stream;
    // SURVEY.md 8d bytes of the fp32 backward
    LaunchScope scope("msda_bwd_f32", 4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D), st);
    // d(loc), d(attn): the 8-lane gather kernel of msda.hip (exact fp32 dot products)
    if (int rc = msda_grad_taps_f32(value, shapes, lsi, loc, attn, grad_out, N, S, M, D, L, Lq, P, grad_loc, grad_attn, st))
        return rc;
    PlainSrc src{loc, attn, (int)(L * P), grad_loc, grad_attn};
    return run_tiled<float, float, PlainSrc, false>(fn, src, bd, shapes, lsi, N, M, L, Lq, S, value, grad_out, grad_value, ws, st);
}

int vah_msda_fused_backward_tiled(const void *value, int value_dtype, const int64_t *shapes, const int64_t *lsi,
                                  const void *offsets, const void *logits, int param_dtype, int64_t offsets_stride,
                                  int64_t logits_stride, const float *ref,
                                  int64_t ref_levels, const void *grad_out, int64_t N, int64_t S, int64_t M, int64_t D,
                                  int64_t L, int64_t Lq, int64_t P, void *grad_value, int grad_value_dtype,
                                  void *d_offsets, void *d_logits, int grad_param_dtype, int64_t d_offsets_stride,
                                  int64_t d_logits_stride, void *ws, int64_t ws_bytes, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_msda_fused_backward_tiled";
    if (N < 0 || S < 1 || M < 1 || Lq < 0 || M * D >= (1LL << 31)) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (D != kD) return fail(VAH_E_UNSUPPORTED, "%s: needs D == 32", fn);
    if (ref_levels != 1 && ref_levels != L) return fail(VAH_E_SHAPE, "%s: ref_levels must be 1 or L", fn);
    if (N * Lq * M == 0) return VAH_OK;
    if (!value || !shapes || !lsi || !offsets || !logits || !ref || !grad_out || !grad_value || !d_offsets || !d_logits || !ws)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)grad_out | (uintptr_t)grad_value | (uintptr_t)ws | (uintptr_t)offsets) % 16 || ((uintptr_t)d_offsets | (uintptr_t)ref) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned", fn);
    if ((value_dtype | param_dtype | grad_value_dtype | grad_param_dtype) & ~1)
        return fail(VAH_E_UNSUPPORTED, "%s: dtype codes must be 0 (f32) or 1 (bf16)", fn);
    if (grad_param_dtype != param_dtype && !(param_dtype == 0 && grad_param_dtype == 1))
        return fail(VAH_E_UNSUPPORTED, "%s: gradients of fp32 offsets / logits may be bf16, not the reverse", fn);
    const int64_t vs = value_dtype ? 2 : 4, ps = param_dtype ? 2 : 4, gps = grad_param_dtype ? 2 : 4, gs = grad_value_dtype ? 2 : 4;
    const int64_t os = offsets_stride ? offsets_stride : L * P * 2, ls = logits_stride ? logits_stride : L * P;
    const int64_t dos = d_offsets_stride ? d_offsets_stride : L * P * 2, dls = d_logits_stride ? d_logits_stride : L * P;
    const bool strided = os != L * P * 2 || ls != L * P || dos != L * P * 2 || dls != L * P;
    if (os < L * P * 2 || ls < L * P || dos < L * P * 2 || dls < L * P || (os * ps) % 16 || (ls * ps) % 8 || (dos * gps) % (gps == 4 ? 16 : 8) ||
        ((uintptr_t)logits) % 8 || (dls * gps) % gps || N * Lq * M * (os > ls ? os : ls) >= ((int64_t)1 << 31))
        return fail(VAH_E_ALIGN, "%s: bad strides", fn);
    Bounds bd;
    if (int rc = make_bounds(fn, N, S, M, L, Lq, P, &bd)) return rc;
    if (ws_bytes < bd.total) return fail(VAH_E_SHAPE, "%s: workspace too small (%lld < %lld)", fn, (long long)ws_bytes, (long long)bd.total);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("msda_fused_bwd", vs * (N * S * M * D + N * Lq * M * D) + gs * N * S * M * D + (ps + gps) * 3 * N * Lq * M * L * P, st,
                      4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D));
    // d(offsets), d(logits) from the gather kernel of msda_fused.hip (nothing scattered) where the tile pass does not
    // compute them itself (fp32 values)
    if (value_dtype != 1) {
        if (grad_param_dtype != param_dtype || strided)
            return fail(VAH_E_UNSUPPORTED, "%s: fp32 values write gradients in the parameter dtype, contiguous tensors only", fn);
        if (int rc = msda_fused_grad_taps(value, value_dtype, shapes, lsi, offsets, logits, param_dtype, ref, ref_levels, grad_out,
                                          N, S, M, L, Lq, P, d_offsets, d_logits, st))
            return rc;
    }
#define VAH_CASE(VT, VC, PT, PC, GPT, GC, LL)                                                                              \
    if (value_dtype == VC && param_dtype == PC && grad_param_dtype == GC && L == LL)                                      \
        return fused_tiled<VT, PT, GPT, LL>(fn, bd, value, shapes, lsi, offsets, logits, os, ls, ref, (int)ref_levels, N, M, Lq, S, \
                                            grad_out, grad_value, grad_value_dtype, d_offsets, d_logits, dos, dls, ws, st)
#define VAH_CASES(LL)                                  \
    VAH_CASE(float, 0, float, 0, float, 0, LL);        \
    VAH_CASE(__bf16, 1, __bf16, 1, __bf16, 1, LL);     \
    VAH_CASE(__bf16, 1, float, 0, float, 0, LL);       \
    VAH_CASE(__bf16, 1, float, 0, __bf16, 1, LL);      \
    VAH_CASE(float, 0, __bf16, 1, __bf16, 1, LL)
    VAH_CASES(1);
    VAH_CASES(3);
    VAH_CASES(4);
#undef VAH_CASES
#undef VAH_CASE
    return fail(VAH_E_UNSUPPORTED, "%s: L = %lld not instantiated", fn, (long long)L);
}

}  // extern "C"
