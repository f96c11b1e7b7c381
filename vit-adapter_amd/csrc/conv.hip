// 3x3 convolutions of the spatial prior module as implicit GEMMs on the matrix cores, NHWC bf16, fp32 accumulate.
//
// Reference: the conv -> SyncBatchNorm -> ReLU stack of SpatialPriorModule
// (/root/reference/detection/mmdet_custom/models/backbones/adapter_modules.py:217-260): 3 -> 64 (stride 2), 64 -> 64,
// 64 -> 64, [max-pool], 64 -> 128 (s2), 128 -> 256 (s2), 256 -> 256 (s2), all 3 x 3, padding 1, no bias.
//
// ONE gather-GEMM kernel serves the forward convolution and the input gradient:
//     out[n][oy*OS + oy0][ox*OS + ox0][co] = sum_t sum_c  W[co][t][c] * in[n][oy*S + ty[t]][ox*S + tx[t]][c]
// (zero outside the input).  Forward, stride s: S = s, 9 taps (dy - 1, dx - 1), OS = 1.  Input gradient, stride 1:
// in = dY, taps (1 - dy, 1 - dx), W re-laid [ci][t][co].  Input gradient, stride 2: one launch per output parity
// (a, b): S = 1 on the dY grid, 1 / 2 / 2 / 4 taps with offsets in {0, +1}, OS = 2, (oy0, ox0) = (a, b).
// Layout of the product: D^T[co][pixel] = W[co][k] X^T[k][pixel], k = (tap, channel): the A operand (weights) and
// the B operand (8 consecutive channels of one pixel) are both plain 16-byte LDS reads of row-major tiles, for any
// tap shift and stride; a lane of the accumulator holds one pixel and 4 consecutive output channels per register
// group, i.e. 8-byte NHWC stores.
// Workgroup: 4 waves, 64 output channels x (WY rows of 32 pixels); weights of a 64-channel input stay in LDS while
// the workgroup walks its pixel tiles (persistent grid), wider inputs are walked in chunks of CK channels.
//
// The weight gradient is its own kernel (conv_wgrad_kernel): dW[co][t][c] = sum_pixels dY[p][co] X[p*S + tap][c],
// a GEMM whose reduction runs over pixels; both operands come from the row-major tiles by transposed LDS reads
// (ds_read_b64_tr_b16).  Per-workgroup partial sums go to a workspace and are summed in a fixed order (no atomics).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "attn_common.h"
#include "common.h"

namespace vah {
namespace {

using attn::bf16x4;
using attn::bf16x8;
using attn::crow;
using attn::f32x16;
using attn::mfma;
using attn::zero16;

constexpr int kMaxTaps = 9;
constexpr int kCoutTile = 64;
constexpr int kTX = 32;                 // pixels per wave row
constexpr int kMaxGroups = 4;

// One tap group = one launch slice (blockIdx.z): the forward convolution and the stride-1 input gradient have one
// group of 9 taps, the stride-2 input gradient one group per output parity (1 / 2 / 2 / 4 taps).
struct TapGroup {
    int T, oy0, ox0, ny, nx;            // taps; first output position and number of positions (per image)
    int tymin, txmin, HY, HX;           // halo tile: rows / columns of input pixels staged per output tile
    int tiles_y, tiles_x, ntiles;
    int ty[kMaxTaps], tx[kMaxTaps], wi[kMaxTaps];      // offsets and index into the 9-tap weight rows
};
struct TapGeom {
    int S, OS, ngroups;
    int N, IH, IW, Cin, Cout, OH, OW, WT;              // WT: taps per weight row (w is (Cout, WT, Cin))
    TapGroup grp[kMaxGroups];
};

// LDS row strides (bytes): + 16 keeps the 16-byte reads of 16 consecutive rows on distinct banks
__host__ __device__ constexpr int px_stride(int CK) { return CK * 2 + 16; }
__host__ __device__ constexpr int w_stride(int CK, int T) { return T * CK * 2 + 16; }

// CK channels per chunk, WY x WC waves (pixel rows x 32-channel output blocks), MAXP staged pieces per thread,
// TMAX = unrolled tap slots
template <int CK, int WY, int WC, int MAXP, int TMAX>
__global__ __launch_bounds__(64 * WY * WC) void conv_taps_kernel(const __bf16 *__restrict__ in, const __bf16 *__restrict__ w,
                                                                 __bf16 *__restrict__ out, TapGeom g) {
    constexpr int NT = 64 * WY * WC;
    constexpr int NCB = 2 / WC;          // 32-channel blocks per wave
    constexpr int PXS = px_stride(CK);
    constexpr int C8 = CK / 8;           // 16-byte pieces per pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TapGroup &gr = g.grp[blockIdx.z];
    const int T = gr.T;
    const int WROW = w_stride(CK, T);
    unsigned char *s_w = smem, *s_x = smem + kCoutTile * WROW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int wy = wave % WY, wc = wave / WY;
    const int co0 = blockIdx.y * kCoutTile;
    const int nchunks = g.Cin / CK;
    const int npieces = gr.HY * gr.HX * C8;

    // piece -> (halo row, halo col, 8-channel group): independent of the tile
    int p_lds[MAXP], p_yx[MAXP], p_goff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int p = threadIdx.x + NT * i;
        const int px = p / C8, c8 = p - px * C8;
        const int hy = px / gr.HX, hx = px - hy * gr.HX;
        p_lds[i] = p < npieces ? px * PXS + c8 * 16 : -1;
        p_yx[i] = (hy << 16) | hx;
        p_goff[i] = (hy * g.IW + hx) * g.Cin + c8 * 8;
    }
    auto stage_w = [&](int chunk) {
        const int per_row = T * C8;                   // 16-byte pieces per output channel
        for (int p = threadIdx.x; p < kCoutTile * per_row; p += NT) {
            const int co = p / per_row, q = p - co * per_row;
            const int t = q / C8, c8 = q - t * C8;
            const bf16x8 x = *reinterpret_cast<const bf16x8 *>(w + ((int64_t)(co0 + co) * g.WT + gr.wi[t]) * g.Cin + chunk * CK + c8 * 8);
            *reinterpret_cast<bf16x8 *>(s_w + co * WROW + (t * CK + c8 * 8) * 2) = x;
        }
    };
    bf16x8 xp[MAXP];
    auto fetch = [&](int tile, int chunk) {
        const int n = tile / (gr.tiles_y * gr.tiles_x), tr = tile - n * gr.tiles_y * gr.tiles_x;
        const int tyi = tr / gr.tiles_x, txi = tr - tyi * gr.tiles_x;
        const int iy0 = tyi * WY * g.S + gr.tymin, ix0 = txi * kTX * g.S + gr.txmin;
        const __bf16 *base = in + (((int64_t)n * g.IH + iy0) * g.IW + ix0) * g.Cin + chunk * CK;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int iy = iy0 + (p_yx[i] >> 16), ix = ix0 + (p_yx[i] & 0xffff);
            const bool ok = p_lds[i] >= 0 && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
            const bf16x8 x = *reinterpret_cast<const bf16x8 *>(ok ? base + p_goff[i] : in);
#pragma unroll
            for (int j = 0; j < 8; ++j) xp[i][j] = ok ? x[j] : (__bf16)0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < MAXP; ++i)
            if (p_lds[i] >= 0) *reinterpret_cast<bf16x8 *>(s_x + p_lds[i]) = xp[i];
    };
    // per-tap LDS offsets of this lane's pixel (B operand) - the weights' are compile-time multiples of CK
    int xoff[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        xoff[t] = t < T ? ((wy * g.S + gr.ty[t] - gr.tymin) * gr.HX + r * g.S + gr.tx[t] - gr.txmin) * PXS + hf * 16 : 0;
    const unsigned char *wb = s_w + (wc * NCB * 32 + r) * WROW + hf * 16;

    if (nchunks == 1) stage_w(0);
    int tile = blockIdx.x;
    if (tile < gr.ntiles) fetch(tile, 0);
    for (; tile < gr.ntiles; tile += gridDim.x) {
        f32x16 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = zero16();
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            __syncthreads();                                  // everyone is done with the previous tile / chunk
            commit();
            if (nchunks > 1) stage_w(chunk);
            __syncthreads();
            // next pieces on their way while this chunk is multiplied
            if (chunk + 1 < nchunks) fetch(tile, chunk + 1);
            else if (tile + (int)gridDim.x < gr.ntiles) fetch(tile + gridDim.x, 0);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t < T) {
#pragma unroll
                    for (int ks = 0; ks < CK / 16; ++ks) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(s_x + xoff[t] + ks * 32);
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
                            acc[cb] = mfma(*reinterpret_cast<const bf16x8 *>(wb + cb * 32 * WROW + (t * CK + ks * 16) * 2), b, acc[cb]);
                    }
                }
            }
        }
        const int n = tile / (gr.tiles_y * gr.tiles_x), tr = tile - n * gr.tiles_y * gr.tiles_x;
        const int tyi = tr / gr.tiles_x, txi = tr - tyi * gr.tiles_x;
        const int oy = tyi * WY + wy, ox = txi * kTX + r;
        if (oy < gr.ny && ox < gr.nx) {
            __bf16 *op = out + (((int64_t)n * g.OH + oy * g.OS + gr.oy0) * g.OW + ox * g.OS + gr.ox0) * g.Cout + co0 + wc * NCB * 32;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (__bf16)acc[cb][4 * q + j];
                    *reinterpret_cast<bf16x4 *>(op + cb * 32 + 8 * q + 4 * hf) = v;
                }
        }
    }
}

template <int CK, int WY, int WC, int MAXP, int TMAX>
int launch_taps(const void *in, const void *w, void *out, TapGeom &g, hipStream_t st) {
    constexpr int NT = 64 * WY * WC;
    int lds = 0, ntiles = 0;
    for (int i = 0; i < g.ngroups; ++i) {
        TapGroup &gr = g.grp[i];
        gr.HY = (WY - 1) * g.S + gr.HY;             // HY / HX arrive as the tap extents (max - min + 1)
        gr.HX = (kTX - 1) * g.S + gr.HX;
        gr.tiles_y = (gr.ny + WY - 1) / WY;
        gr.tiles_x = (gr.nx + kTX - 1) / kTX;
        gr.ntiles = g.N * gr.tiles_y * gr.tiles_x;
        const int l = kCoutTile * w_stride(CK, gr.T) + gr.HY * gr.HX * px_stride(CK);
        lds = l > lds ? l : lds;
        ntiles = gr.ntiles > ntiles ? gr.ntiles : ntiles;
        if (gr.HY * gr.HX * (CK / 8) > NT * MAXP || l > 160 * 1024 || gr.T > TMAX)
            return fail(VAH_E_SHAPE, "conv_taps: halo tile %d x %d x %d channels / %d taps does not fit", gr.HY, gr.HX, CK, gr.T);
    }
    if (ntiles == 0) return VAH_OK;
    if (int rc = allow_dynamic_lds((const void *)conv_taps_kernel<CK, WY, WC, MAXP, TMAX>, lds, "conv_taps")) return rc;
    // persistent over pixel tiles: the weights of a one-chunk input are staged once per workgroup
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    const int others = (g.Cout / kCoutTile) * g.ngroups;
    int slots = (per_cu * kCUs + others - 1) / others;
    slots = slots < 1 ? 1 : (slots > ntiles ? ntiles : slots);
    hipLaunchKernelGGL((conv_taps_kernel<CK, WY, WC, MAXP, TMAX>), dim3(slots, g.Cout / kCoutTile, g.ngroups), dim3(NT), lds, st,
                       (const __bf16 *)in, (const __bf16 *)w, (__bf16 *)out, g);
    return check_launch("conv_taps");
}

// fills the extents of a group whose taps are set
int finish_group(TapGroup &gr) {
    int ymin = gr.ty[0], ymax = gr.ty[0], xmin = gr.tx[0], xmax = gr.tx[0];
    for (int t = 0; t < gr.T; ++t) {
        if (gr.ty[t] < -4 || gr.ty[t] > 4 || gr.tx[t] < -4 || gr.tx[t] > 4) return -1;
        ymin = gr.ty[t] < ymin ? gr.ty[t] : ymin, ymax = gr.ty[t] > ymax ? gr.ty[t] : ymax;
        xmin = gr.tx[t] < xmin ? gr.tx[t] : xmin, xmax = gr.tx[t] > xmax ? gr.tx[t] : xmax;
    }
    gr.tymin = ymin, gr.txmin = xmin, gr.HY = ymax - ymin + 1, gr.HX = xmax - xmin + 1;
    return 0;
}

int dispatch_taps(const void *in, const void *w, void *out, TapGeom &g, hipStream_t st) {
    if (g.Cin == 16) return g.S == 1 ? launch_taps<16, 8, 1, 2, 9>(in, w, out, g, st) : launch_taps<16, 2, 2, 3, 9>(in, w, out, g, st);
    if (g.S == 2) return launch_taps<64, 2, 2, 11, 9>(in, w, out, g, st);
    // 8 rows of 32 pixels per workgroup: two waves per SIMD behind one copy of the weights
    int tmax = 0;
    for (int i = 0; i < g.ngroups; ++i) tmax = g.grp[i].T > tmax ? g.grp[i].T : tmax;
    return tmax > 4 ? launch_taps<64, 8, 1, 6, 9>(in, w, out, g, st) : launch_taps<64, 8, 1, 6, 4>(in, w, out, g, st);
}

// ---------------------------------------------------------------------------------------
// Weight gradient: dW[co][t][c] = sum over pixels p of dY[p][co] * X[p*S + tap t][c]   (3 x 3, padding 1).
// Workgroup = (pixel-tile slot, 64 output channels, CK input channels): 2 x 9 x (CK / 32) blocks of 32 x 32, split
// over the 4 waves; the reduction index of the MFMA is 16 consecutive pixels of one tile row:
//   A = dY^T[m = co][k = pixel]  from the row-major dY tile [pixel][co]  by ds_read_b64_tr_b16,
//   B = X  [k = pixel][n = c]   from the row-major X halo tile [pixel][c] by ds_read_b64_tr_b16 (the tap shifts the
//       ROW, so every read stays 8-byte aligned).
// Every workgroup adds up its tiles in registers and writes one fp32 partial (64 x 9 x CK) to the workspace;
// conv_wgrad_reduce sums the partials of a (co tile, c tile) in slot order: bitwise reproducible, no atomics.
// ---------------------------------------------------------------------------------------
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;

struct WgradGeom {
    int S, N, IH, IW, Cin, OH, OW, Cout;
    int HY, HX, tiles_y, tiles_x, ntiles;
};

// transposed fragment: rows row0 + 4 * hf + q (q = 0..3) and 8 rows further, columns col0 + 16 * (grp & 1) + 4 p ..;
// the lane receives column (lane & 31) of the 8 rows in the k order of a standard fragment pair (4 hf.., 8 + 4 hf..)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *p0, const unsigned char *p1) {
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p0);
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int CK, int WY>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy,
                                                         float *__restrict__ part, WgradGeom g) {
    constexpr int NT = 512, HW = 4;                                  // 8 waves: HW per 32-channel output block
    constexpr int PXS = px_stride(CK), PYS = px_stride(kCoutTile);
    constexpr int C8 = CK / 8, NCI = CK / 32 > 0 ? CK / 32 : 1;      // 32-wide input-channel blocks (CK = 16: one, half used)
    constexpr int PER = (9 * NCI + HW - 1) / HW;                     // blocks per wave; waves HW cob .. HW cob + 3 share output block cob
    constexpr int NPIX = WY * kTX;
    constexpr int MAXP = CK == 64 ? (WY == 4 ? 4 : 6) : 2;           // X pieces per thread (host checks)
    constexpr int DYP = (NPIX * 8 + NT - 1) / NT;                    // dY pieces per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_dy = smem, *s_x = smem + NPIX * PYS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, i16 = lane & 15, hf = lane >> 5;
    const int co0 = blockIdx.y * kCoutTile, c0 = blockIdx.z * CK;
    const int npieces = g.HY * g.HX * C8;
    int p_lds[MAXP], p_yx[MAXP], p_goff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int p = threadIdx.x + NT * i;
        const int px = p / C8, c8 = p - px * C8;
        const int hy = px / g.HX, hx = px - hy * g.HX;
        p_lds[i] = p < npieces ? px * PXS + c8 * 16 : -1;
        p_yx[i] = (hy << 16) | hx;
        p_goff[i] = (hy * g.IW + hx) * g.Cin + c8 * 8;
    }
    bf16x8 xp[MAXP], yp[DYP];
    auto fetch = [&](int tile) {
        const int n = tile / (g.tiles_y * g.tiles_x), tr = tile - n * g.tiles_y * g.tiles_x;
        const int tyi = tr / g.tiles_x, txi = tr - tyi * g.tiles_x;
        const int oy0 = tyi * WY, ox0 = txi * kTX;
        const int iy0 = oy0 * g.S - 1, ix0 = ox0 * g.S - 1;
        const __bf16 *xb = x + (((int64_t)n * g.IH + iy0) * g.IW + ix0) * g.Cin + c0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int iy = iy0 + (p_yx[i] >> 16), ix = ix0 + (p_yx[i] & 0xffff);
            const bool ok = p_lds[i] >= 0 && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(ok ? xb + p_goff[i] : x);
#pragma unroll
            for (int j = 0; j < 8; ++j) xp[i][j] = ok ? v[j] : (__bf16)0.f;
        }
#pragma unroll
        for (int i = 0; i < DYP; ++i) {
            const int p = threadIdx.x + NT * i, px = p >> 3, c8 = p & 7;
            const int oy = oy0 + px / kTX, ox = ox0 + px % kTX;
            const bool ok = px < NPIX && oy < g.OH && ox < g.OW;
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(ok ? dy + (((int64_t)n * g.OH + oy) * g.OW + ox) * g.Cout + co0 + c8 * 8 : dy);
#pragma unroll
            for (int j = 0; j < 8; ++j) yp[i][j] = ok ? v[j] : (__bf16)0.f;
        }
    };
    f32x16 acc[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) acc[j] = zero16();
    const int cob = wave / HW, sub = wave % HW;

    int tile = blockIdx.x;
    if (tile < g.ntiles) fetch(tile);
    for (; tile < g.ntiles; tile += gridDim.x) {
        __syncthreads();                                       // previous tile's reads are done
#pragma unroll
        for (int i = 0; i < MAXP; ++i)
            if (p_lds[i] >= 0) *reinterpret_cast<bf16x8 *>(s_x + p_lds[i]) = xp[i];
#pragma unroll
        for (int i = 0; i < DYP; ++i) {
            const int p = threadIdx.x + NT * i;
            if (p < NPIX * 8) *reinterpret_cast<bf16x8 *>(s_dy + (p >> 3) * PYS + (p & 7) * 16) = yp[i];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);       // on its way while this tile is multiplied
        // the blocks of a wave share one output-channel block: its dY^T fragment is read once per 16-pixel step
#pragma unroll 2
        for (int ks = 0; ks < NPIX / 16; ++ks) {               // 16 consecutive pixels of one tile row
            const int row = ks / (kTX / 16), col0 = (ks % (kTX / 16)) * 16;
            const int pa = row * kTX + col0 + 4 * hf + (i16 >> 2);                       // dY pixel of this lane's address
            const unsigned char *a0 = s_dy + pa * PYS + (cob * 32 + 16 * (grp & 1) + 4 * (i16 & 3)) * 2;
            const bf16x8 a = tr_frag(a0, a0 + 8 * PYS);
            const unsigned char *bx = s_x + ((row * g.S) * g.HX + (col0 + 4 * hf + (i16 >> 2)) * g.S) * PXS + (16 * (grp & 1) + 4 * (i16 & 3)) * 2;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int rem = sub * PER + j;
                if (rem < 9 * NCI) {
                    const int t = rem / NCI, cib = rem - t * NCI;
                    const int dyo = t / 3, dxo = t - 3 * dyo;
                    const unsigned char *b0 = bx + (dyo * g.HX + dxo) * PXS + cib * 64;
                    acc[j] = mfma(a, tr_frag(b0, b0 + 8 * g.S * PXS), acc[j]);
                }
            }
        }
    }
    // partial [slot][co tile][c tile] -> (64 co) x 9 x CK floats; lane holds column c = lane & 31, rows co = crow
    float *pp = part + (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (kCoutTile * 9 * CK);
    const int r = lane & 31;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int rem = sub * PER + j;
        if (rem < 9 * NCI) {
            const int t = rem / NCI, cib = rem - t * NCI;
            if (cib * 32 + r < CK) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pp[((cob * 32 + crow(i, hf)) * 9 + t) * CK + cib * 32 + r] = acc[j][i];
            }
        }
    }
}

// dW[co][t][c] (fp32, Cout x 9 x Cin) = sum over slots of the partials: 4 threads per element, each with 4 independent
// running sums over its slots (a single dependent chain of `slots` loads is latency bound), combined in a fixed order
__global__ __launch_bounds__(256) void conv_wgrad_reduce(const float *__restrict__ part, int slots, int ncot, int ncit, int CK,
                                                         int Cin, float *__restrict__ dw, int64_t total) {
    __shared__ float s_sum[4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < total) {
        const int c = (int)(i % Cin), t = (int)(i / Cin % 9), co = (int)(i / (9 * Cin));
        const int cot = co / kCoutTile, cit = c / CK;
        const int64_t sz = (int64_t)kCoutTile * 9 * CK, stride = (int64_t)ncot * ncit * sz;
        const float *p = part + ((int64_t)cot * ncit + cit) * sz + ((int64_t)(co % kCoutTile) * 9 + t) * CK + c % CK;
        int sl = q;
        for (; sl + 12 < slots; sl += 16) {
            s[0] += p[sl * stride];
            s[1] += p[(sl + 4) * stride];
            s[2] += p[(sl + 8) * stride];
            s[3] += p[(sl + 12) * stride];
        }
        for (; sl < slots; sl += 4) s[0] += p[sl * stride];
    }
    s_sum[q][e] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (q == 0 && i < total) dw[i] = (s_sum[0][e] + s_sum[1][e]) + (s_sum[2][e] + s_sum[3][e]);
}

template <int CK, int WY>
int launch_wgrad(const void *x, const void *dy, float *ws, int64_t ws_floats, float *dw, WgradGeom &g, hipStream_t st) {
    g.HY = (WY - 1) * g.S + 3;
    g.HX = (kTX - 1) * g.S + 3;
    g.tiles_y = (g.OH + WY - 1) / WY;
    g.tiles_x = (g.OW + kTX - 1) / kTX;
    g.ntiles = g.N * g.tiles_y * g.tiles_x;
    const int lds = WY * kTX * px_stride(kCoutTile) + g.HY * g.HX * px_stride(CK) + 64;   // + 64: a 16-channel tile is read 32 wide
    const int ncot = g.Cout / kCoutTile, ncit = g.Cin / CK;
    int slots = (kCUs + ncot * ncit - 1) / (ncot * ncit);                 // one resident workgroup per CU
    if (slots > g.ntiles) slots = g.ntiles;
    const int64_t psz = (int64_t)kCoutTile * 9 * CK;
    while (slots > 1 && (int64_t)slots * ncot * ncit * psz > ws_floats) --slots;
    if ((int64_t)slots * ncot * ncit * psz > ws_floats) return fail(VAH_E_SHAPE, "vah_conv3x3_wgrad_nhwc_bf16: workspace too small");
    if (int rc = allow_dynamic_lds((const void *)conv_wgrad_kernel<CK, WY>, lds, "conv_wgrad")) return rc;
    hipLaunchKernelGGL((conv_wgrad_kernel<CK, WY>), dim3(slots, ncot, ncit), dim3(512), lds, st, (const __bf16 *)x,
                       (const __bf16 *)dy, ws, g);
    if (int rc = check_launch("conv_wgrad")) return rc;
    const int64_t total = (int64_t)g.Cout * 9 * g.Cin;
    hipLaunchKernelGGL(conv_wgrad_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float *)ws, slots, ncot,
                       ncit, CK, g.Cin, dw, total);
    return check_launch("conv_wgrad_reduce");
}

}  // namespace
}  // namespace vah

extern "C" {

int vah_conv_taps_nhwc_bf16(const void *in, int64_t N, int64_t IH, int64_t IW, int64_t Cin, const void *w, int64_t Cout,
                            int T, const int *ty, const int *tx, int S, void *out, int64_t ny, int64_t nx, int64_t OH,
                            int64_t OW, int OS, int oy0, int ox0, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_conv_taps_nhwc_bf16";
    if (T < 1 || T > kMaxTaps || (S != 1 && S != 2) || (OS != 1 && OS != 2) || !ty || !tx)
        return fail(VAH_E_SHAPE, "%s: 1 <= taps <= 9, strides 1 or 2", fn);
    if (N < 0 || IH < 1 || IW < 1 || ny < 0 || nx < 0 || OH < 1 || OW < 1 || (Cin != 16 && Cin % 64) || Cin < 16 || Cout < 64 ||
        Cout % 64 || IH > 32767 || IW > 32767 || N * IH * IW * Cin >= (1ll << 31) || N * OH * OW * Cout >= (1ll << 31))
        return fail(VAH_E_SHAPE, "%s: Cin must be 16 or a multiple of 64, Cout a multiple of 64; tensors < 2^31 elements", fn);
    if ((ny - 1) * OS + oy0 >= OH || (nx - 1) * OS + ox0 >= OW || oy0 < 0 || ox0 < 0)
        return fail(VAH_E_SHAPE, "%s: output positions leave the output tensor", fn);
    if (N == 0 || ny == 0 || nx == 0) return VAH_OK;
    if (!in || !w || !out) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)in | (uintptr_t)w) % 16 || (uintptr_t)out % 8) return fail(VAH_E_ALIGN, "%s: misaligned operand", fn);
    TapGeom g{};
    g.S = S, g.OS = OS, g.ngroups = 1, g.WT = T;
    g.N = (int)N, g.IH = (int)IH, g.IW = (int)IW, g.Cin = (int)Cin, g.Cout = (int)Cout, g.OH = (int)OH, g.OW = (int)OW;
    TapGroup &gr = g.grp[0];
    gr.T = T, gr.oy0 = oy0, gr.ox0 = ox0, gr.ny = (int)ny, gr.nx = (int)nx;
    for (int t = 0; t < T; ++t) gr.ty[t] = ty[t], gr.tx[t] = tx[t], gr.wi[t] = t;
    if (finish_group(gr)) return fail(VAH_E_SHAPE, "%s: tap offsets within +-4", fn);
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes: input and output once, weights once; flops: 2 * outputs * taps * Cin
    LaunchScope scope("conv_taps_bf16", (N * IH * IW * Cin + N * ny * nx * Cout + Cout * T * Cin) * 2, st, 0,
                      2 * N * ny * nx * Cout * T * Cin);
    return dispatch_taps(in, w, out, g, st);
}

int vah_conv3x3_dgrad_nhwc_bf16(const void *gy, int64_t N, int64_t OH, int64_t OW, int64_t Cout, const void *wt, int64_t Cin,
                                int S, void *gx, int64_t H, int64_t W, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_conv3x3_dgrad_nhwc_bf16";
    if ((S != 1 && S != 2) || N < 0 || H < 1 || W < 1 || OH != (H - 1) / S + 1 || OW != (W - 1) / S + 1 || Cout % 64 || Cout < 64 ||
        Cin % 64 || Cin < 64 || H > 32767 || W > 32767 || N * H * W * Cin >= (1ll << 31) || N * OH * OW * Cout >= (1ll << 31))
        return fail(VAH_E_SHAPE, "%s: 3x3 / padding 1 / stride 1 or 2 shapes; Cin, Cout multiples of 64", fn);
    if (N == 0) return VAH_OK;
    if (!gy || !wt || !gx) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)gy | (uintptr_t)wt) % 16 || (uintptr_t)gx % 8) return fail(VAH_E_ALIGN, "%s: misaligned operand", fn);
    // the gather runs over dY: "input" = dY (Cout channels), "output" = dX (Cin channels), weights (Cin, 9, Cout)
    TapGeom g{};
    g.S = 1, g.OS = S, g.WT = 9;
    g.N = (int)N, g.IH = (int)OH, g.IW = (int)OW, g.Cin = (int)Cout, g.Cout = (int)Cin, g.OH = (int)H, g.OW = (int)W;
    if (S == 1) {
        // dx[y][x] = sum_{dy,dx} gy[y + 1 - dy][x + 1 - dx] w[:, :, dy, dx]
        g.ngroups = 1;
        TapGroup &gr = g.grp[0];
        gr.T = 9, gr.oy0 = gr.ox0 = 0, gr.ny = (int)H, gr.nx = (int)W;
        for (int t = 0; t < 9; ++t) gr.ty[t] = 1 - t / 3, gr.tx[t] = 1 - t % 3, gr.wi[t] = t;
        finish_group(gr);
    } else {
        // output parity (a, b): y = 2j + a receives gy[j + (a + 1 - dy) / 2] for the dy with a + 1 - dy even
        g.ngroups = 0;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                TapGroup gr{};
                gr.oy0 = a, gr.ox0 = b, gr.ny = (int)((H - a + 1) / 2), gr.nx = (int)((W - b + 1) / 2);
                if (gr.ny <= 0 || gr.nx <= 0) continue;
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx)
                        if ((a + 1 - dy) % 2 == 0 && (b + 1 - dx) % 2 == 0) {
                            gr.ty[gr.T] = (a + 1 - dy) / 2, gr.tx[gr.T] = (b + 1 - dx) / 2, gr.wi[gr.T] = dy * 3 + dx;
                            ++gr.T;
                        }
                finish_group(gr);
                g.grp[g.ngroups++] = gr;
            }
    }
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("conv_dgrad_bf16", (N * OH * OW * Cout + N * H * W * Cin + Cout * 9 * Cin) * 2, st, 0,
                      2 * N * OH * OW * Cout * 9 * Cin);
    return dispatch_taps(gy, wt, gx, g, st);
}

int64_t vah_conv3x3_wgrad_ws_floats(int64_t Cin, int64_t Cout) {
    // one workgroup slot per CU in all, each with a 64 x 9 x CK fp32 partial
    const int64_t CK = Cin == 16 ? 16 : 64, pairs = (Cout / 64) * (Cin / CK);
    const int64_t slots = (vah::kCUs + pairs - 1) / pairs;
    return slots * pairs * 64 * 9 * CK;
}

int vah_conv3x3_wgrad_nhwc_bf16(const void *x, int64_t N, int64_t IH, int64_t IW, int64_t Cin, const void *dy, int64_t OH,
                                int64_t OW, int64_t Cout, int S, float *ws, int64_t ws_floats, float *dw, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_conv3x3_wgrad_nhwc_bf16";
    if ((S != 1 && S != 2) || N < 0 || IH < 1 || IW < 1 || (Cin != 16 && Cin % 64) || Cin < 16 || Cout < 64 || Cout % 64 ||
        OH != (IH - 1) / S + 1 || OW != (IW - 1) / S + 1 || IH > 32767 || IW > 32767 || N * IH * IW * Cin >= (1ll << 31) ||
        N * OH * OW * Cout >= (1ll << 31))
        return fail(VAH_E_SHAPE, "%s: 3x3 / padding 1 / stride 1 or 2 shapes; Cin 16 or a multiple of 64, Cout a multiple of 64", fn);
    if (!dw || !ws) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return (int)hipMemsetAsync(dw, 0, Cout * 9 * Cin * 4, st);
    if (!x || !dy) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)x | (uintptr_t)dy) % 16) return fail(VAH_E_ALIGN, "%s: misaligned operand", fn);
    WgradGeom g{};
    g.S = S, g.N = (int)N, g.IH = (int)IH, g.IW = (int)IW, g.Cin = (int)Cin, g.OH = (int)OH, g.OW = (int)OW, g.Cout = (int)Cout;
    LaunchScope scope("conv_wgrad_bf16", (N * IH * IW * Cin + N * OH * OW * Cout) * 2 + Cout * 9 * Cin * 4, st, 0,
                      2 * N * OH * OW * Cout * 9 * Cin);
    if (Cin == 16) return S == 1 ? launch_wgrad<16, 4>(x, dy, ws, ws_floats, dw, g, st) : launch_wgrad<16, 2>(x, dy, ws, ws_floats, dw, g, st);
    return S == 1 ? launch_wgrad<64, 4>(x, dy, ws, ws_floats, dw, g, st) : launch_wgrad<64, 2>(x, dy, ws, ws_floats, dw, g, st);
}

}  // extern "C"
