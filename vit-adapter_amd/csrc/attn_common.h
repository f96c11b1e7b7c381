// Shared pieces of the bf16 MFMA attention kernels (gfx950, head_dim 64).
//
// All kernels use v_mfma_f32_32x32x16_bf16.  Lane l = (r = l & 31, h = l >> 5):
//   A operand element j : A[row r][k = 8h + j]         (8 bf16)
//   B operand element j : B[k = 8h + j][col r]         (8 bf16)
//   C/D register i      : C[row (i&3) + 8(i>>2) + 4h][col r]
// A 32x32 f32 accumulator X (rows on registers, column on the lane) can feed the next MFMA
// without touching LDS when that product sums over X's ROW index: registers 8s..8s+7, converted
// to bf16, are the k-step-s fragment, whose element j is X row 16s + 8(j>>2) + 4h + (j&3); the
// OTHER operand's element j must then come from that same k (see krow()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vah {
namespace attn {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

constexpr int kHD = 64;          // head dim
constexpr int kPadRow = 72;      // LDS row stride (bf16) of a [rows][64] tile read with b128: 144 B
constexpr int kPadT = 68;        // LDS row stride (bf16) of a [64 d][64 keys] transposed tile, b64 reads
constexpr int kPadT32 = 36;      // LDS row stride (bf16) of a [64 d][32 queries] transposed tile

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// row of a C/D tile held in register i by lane half h
__device__ __forceinline__ int crow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// registers 8s..8s+7 of an accumulator -> bf16 fragment of k-step s
__device__ __forceinline__ bf16x8 pack_half(const f32x16 &x, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)x[8 * s + j];
    return f;
}

// Fragment of a TRANSPOSED tile T[row = r][k] whose k order must match pack_half: elements 0..3
// are k = 16s + 4h + 0..3 and elements 4..7 are k = 16s + 8 + 4h + 0..3 (two 8-byte reads).
__device__ __forceinline__ bf16x8 load_kperm(const __bf16 *row_ptr, int s, int h) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(row_ptr + 16 * s + 4 * h);
    const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(row_ptr + 16 * s + 8 + 4 * h);
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = lo[j];
        f[4 + j] = hi[j];
    }
    return f;
}

// Token addressing.  win == 0: sequence z is tokens [z*N, (z+1)*N) of a (B*N, ...) row-major tensor.
// win > 0: sequence z = (image b, window wy, wx) of a (B, H, W) token grid cut into win x win windows
// (grid padded up to a multiple of win); local token i = ly*win + lx maps to pixel (wy*win+ly,
// wx*win+lx), or to NO row (-1) when that pixel lies in the padding: such tokens read as zero rows
// (the reference zero-pads q, k, v AFTER the projection, base/vit.py:143-145) and are never stored.
struct RowMap {
    int win, H, W, nwx, nwin;
};
__device__ __forceinline__ int64_t grow(const RowMap &rm, int z, int i, int N) {
    if (rm.win == 0) return (int64_t)z * N + i;
    const int b = z / rm.nwin, w = z - b * rm.nwin;
    const int wy = w / rm.nwx, wx = w - wy * rm.nwx;
    const int ly = i / rm.win, lx = i - ly * rm.win;
    const int y = wy * rm.win + ly, x = wx * rm.win + lx;
    if (y >= rm.H || x >= rm.W) return -1;
    return ((int64_t)b * rm.H + y) * rm.W + x;
}

// host: window geometry -> RowMap, number of sequences Z and tokens per sequence N (attn_fwd.hip)
int make_rowmap(const char *fn, int64_t win, int64_t B, int64_t gh, int64_t gw, int64_t *Z, int64_t *N,
                RowMap *rm);

// attn_flash.hip: whole-sequence forward / backward (rows b * N + i of a (B * N, ld) tensor); bias (bf16, times
// log2(e), (H, N, ldb)) and its transpose are optional, ds_out (B, H, N, ldb) receives d loss / d bias per image
int attn_fwd_seq(const void *q, const void *k, const void *v, int64_t ld, int64_t B, int64_t H, int64_t N, float scale,
                 const void *bias, int64_t ldb, void *out, int64_t ld_out, float *lse, hipStream_t st);
int attn_bwd_seq(const void *q, const void *k, const void *v, int64_t ld, const void *o, const void *d_o, int64_t ld_out,
                 const float *lse, int64_t B, int64_t H, int64_t N, float scale, const void *bias, const void *bias_t,
                 int64_t ldb, void *ds_out, float *delta, void *dq, void *dk, void *dv, int64_t ld_d, hipStream_t st);

// attn_win.hip: forward with the whole window (N <= 224 tokens) resident in LDS, one workgroup per (window, head)
int attn_win_fwd_resident(const void *q, const void *k, const void *v, int64_t ld, RowMap rm, int64_t Z, int64_t H,
                          int64_t N, float scale, void *out, int64_t ld_out, float *lse, hipStream_t st);

int attn_win_bwd_resident(const void *q, const void *k, const void *v, int64_t ld, const void *o, const void *d_o,
                          int64_t ld_out, RowMap rm, const float *lse, int64_t Z, int64_t H, int64_t N, float scale, void *dq,
                          void *dk, void *dv, int64_t ld_d, hipStream_t st);

// (rows, heads, 64) strided -> (Z, heads, 64, Np) dense, zero padded beyond N / outside the image.
// One 64-token tile; `tile` is 64 * kPadRow bf16 of LDS, left holding the (token, d) tile.
__device__ __forceinline__ void transpose_tile_to_dn(const __bf16 *__restrict__ src, int64_t ld, const RowMap &rm,
                                                     int N, int Np, int H, __bf16 *__restrict__ dst, int n0, int h,
                                                     int b, __bf16 *tile) {
    const __bf16 *s = src + (int64_t)h * kHD;
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int row = c >> 3, col = (c & 7) * 8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
        const int64_t gr = n0 + row < N ? grow(rm, b, n0 + row, N) : -1;
        if (gr >= 0) v = *reinterpret_cast<const bf16x8 *>(s + gr * ld + col);
        *reinterpret_cast<bf16x8 *>(tile + row * kPadRow + col) = v;
    }
    __syncthreads();
    __bf16 *d = dst + ((int64_t)(b * H + h) * kHD) * Np + n0;
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int drow = c >> 3, k0 = (c & 7) * 8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[(k0 + j) * kPadRow + drow];
        *reinterpret_cast<bf16x8 *>(d + (int64_t)drow * Np + k0) = v;
    }
}

static __global__ __launch_bounds__(256) void transpose_to_dn(const __bf16 *__restrict__ src, int64_t ld,
                                                              RowMap rm, int N, int Np, int H,
                                                              __bf16 *__restrict__ dst) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[64 * kPadRow];
    transpose_tile_to_dn(src, ld, rm, N, Np, H, dst, blockIdx.x * 64, blockIdx.y, blockIdx.z, tile);
}

}  // namespace attn
}  // namespace vah
