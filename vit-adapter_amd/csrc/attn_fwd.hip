// Fused softmax attention forward for gfx950, bf16 in / bf16 out, fp32 accumulate, head_dim 64.
//
// Arithmetic of the reference's Attention / WindowedAttention
// (/root/reference/detection/mmdet_custom/models/backbones/base/vit.py:83-88,154-159):
//   out = softmax(q k^T * scale) v   per (batch, head); no mask, no dropout.
// The (N x N) score matrix is never written to memory: a workgroup walks the keys in tiles of
// 64 with an online softmax (running max / sum per query) - the reference materialises
// (B, heads, N, N) scores (1.6 GB fp32 for ViT-B at 1024^2).
//
// Layout / tiling (MI355X):
//   * q, k, v are read IN PLACE from the fused qkv projection (B, N, 3, heads, 64): row stride
//     `ld` elements, no permute / contiguous copies.  out is (B, N, heads, 64).
//   * one workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries.
//   * scores are computed TRANSPOSED, S^T = K Q^T (keys on accumulator registers, the query on the
//     lane), so the softmax statistics of a query live in one lane (+ its partner lane l^32) and
//     the probabilities feed the second product  O^T = V^T P^T  straight from registers
//     (attn_common.h).  V^T comes from a (B, heads, 64, Np) copy made by a small transpose kernel,
//     so both LDS tiles are filled with plain 16-byte copies: K as [64 keys][64 d] read with
//     ds_read_b128 (144-byte rows: conflict free), V^T as [64 d][64 keys] read with ds_read_b64.
//   * the next K / V^T tile is fetched into registers while the current one is being multiplied.
#include "attn_common.h"
#include "common.h"

namespace vah {
namespace attn {
namespace {

constexpr int kWaves = 4;
constexpr int kQBlock = 32 * kWaves;     // queries per workgroup
constexpr int kKTile = 64;               // keys per step

__global__ __launch_bounds__(256) void attn_fwd_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ vt,
    int64_t ld, RowMap rm, int N, int Np, int H, float scale_log2,
    __bf16 *__restrict__ out, int64_t ld_out, float *__restrict__ lse) {
    // double buffered: tile t+1 is written while tile t is being multiplied (one barrier per tile)
    __shared__ __attribute__((aligned(16))) __bf16 s_k2[2][kKTile * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_vt2[2][kHD * kPadT];

    const int h = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int qrow = blockIdx.x * kQBlock + wave * 32 + r;           // this lane's query
    const int qload = min(qrow, N - 1);

    const __bf16 *qb = q + (int64_t)h * kHD;
    const __bf16 *kb = k + (int64_t)h * kHD;
    const __bf16 *vtb = vt + ((int64_t)(b * H + h) * kHD) * Np;
    const int64_t gq = grow(rm, b, qload, N);          // -1: padded token, q = 0

    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[kk][j] = (__bf16)0.f;
        if (gq >= 0) qf[kk] = *reinterpret_cast<const bf16x8 *>(qb + gq * ld + 16 * kk + 8 * hf);
    }

    f32x16 o[2] = {zero16(), zero16()};
    float m_run = -INFINITY, l_run = 0.f;

    // staging: 512 16-byte chunks per tile and per matrix, 2 per thread
    const int c0 = threadIdx.x, c1 = threadIdx.x + 256;
    const int kr0 = c0 >> 3, kc0 = (c0 & 7) * 8, kr1 = c1 >> 3, kc1 = (c1 & 7) * 8;
    bf16x8 pk0, pk1, pv0, pv1;
    auto fetch = [&](int key0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pk0[j] = pk1[j] = (__bf16)0.f;
        const int64_t g0 = key0 + kr0 < N ? grow(rm, b, key0 + kr0, N) : -1;
        const int64_t g1 = key0 + kr1 < N ? grow(rm, b, key0 + kr1, N) : -1;
        if (g0 >= 0) pk0 = *reinterpret_cast<const bf16x8 *>(kb + g0 * ld + kc0);
        if (g1 >= 0) pk1 = *reinterpret_cast<const bf16x8 *>(kb + g1 * ld + kc1);
        pv0 = *reinterpret_cast<const bf16x8 *>(vtb + (int64_t)kr0 * Np + key0 + kc0);
        pv1 = *reinterpret_cast<const bf16x8 *>(vtb + (int64_t)kr1 * Np + key0 + kc1);
    };
    auto commit = [&](int buf) {
        __bf16 *s_k = s_k2[buf], *s_vt = s_vt2[buf];
        *reinterpret_cast<bf16x8 *>(s_k + kr0 * kPadRow + kc0) = pk0;
        *reinterpret_cast<bf16x8 *>(s_k + kr1 * kPadRow + kc1) = pk1;
        // V^T rows are 64 keys wide: two 8-byte halves keep 8-byte alignment under the 136-byte stride
        const bf16x4 *a0 = reinterpret_cast<const bf16x4 *>(&pv0), *a1 = reinterpret_cast<const bf16x4 *>(&pv1);
        *reinterpret_cast<bf16x4 *>(s_vt + kr0 * kPadT + kc0) = a0[0];
        *reinterpret_cast<bf16x4 *>(s_vt + kr0 * kPadT + kc0 + 4) = a0[1];
        *reinterpret_cast<bf16x4 *>(s_vt + kr1 * kPadT + kc1) = a1[0];
        *reinterpret_cast<bf16x4 *>(s_vt + kr1 * kPadT + kc1 + 4) = a1[1];
    };

    const int ntiles = (N + kKTile - 1) / kKTile;
    fetch(0);
    commit(0);
    if (ntiles > 1) fetch(kKTile);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                 // tile t visible; every wave is done with tile t-1
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);         // overwrites the buffer tile t-1 lived in
            if (t + 2 < ntiles) fetch((t + 2) * kKTile);
        }
        const __bf16 *s_k = s_k2[t & 1], *s_vt = s_vt2[t & 1];

        // S^T = K Q^T : two blocks of 32 keys
        f32x16 s[2] = {zero16(), zero16()};
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(s_k + (kbk * 32 + r) * kPadRow + 16 * kk + 8 * hf);
                s[kbk] = mfma(a, qf[kk], s[kbk]);
            }
        const int key0 = t * kKTile;
        float mx = -INFINITY;
        if (key0 + kKTile > N) {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + kbk * 32 + crow(i, hf) >= N) s[kbk][i] = -INFINITY;
        }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kbk][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // first tile: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(s[kbk][i] * scale_log2 - m_new);
                s[kbk][i] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][i] *= alpha;

        // O^T += V^T P^T
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack_half(s[kbk], sp);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 a = load_kperm(s_vt + (db * 32 + r) * kPadT + kbk * 32, sp, hf);
                    o[db] = mfma(a, pf, o[db]);
                }
            }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow < N && gq >= 0) {
        __bf16 *op = out + gq * ld_out + (int64_t)h * kHD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (__bf16)(o[db][4 * g + j] * inv);
                *reinterpret_cast<bf16x4 *>(op + db * 32 + 8 * g + 4 * hf) = w;
            }
        if (hf == 0) lse[((int64_t)b * H + h) * N + qrow] = m_run + log2f(l_tot);
    } else if (qrow < N && hf == 0) {
        lse[((int64_t)b * H + h) * N + qrow] = INFINITY;     // padded query: p = 0 in the backward
    }
}

}  // namespace
}  // namespace attn
}  // namespace vah

extern "C" {

int64_t vah_attn_padded_len(int64_t N) { return (N + 63) / 64 * 64; }

}  // extern "C"

namespace vah {
namespace attn {
// Shared argument handling of the plain and the windowed entry points: Z sequences of N tokens.
int make_rowmap(const char *fn, int64_t win, int64_t B, int64_t gh, int64_t gw, int64_t *Z, int64_t *N,
                RowMap *rm) {
    if (win == 0) {
        *rm = RowMap{0, 0, 0, 0, 0};
        return VAH_OK;
    }
    if (win < 1 || win > 64 || gh < 1 || gw < 1 || B < 0)
        return fail(VAH_E_SHAPE, "%s: bad window geometry", fn);
    const int64_t nwy = (gh + win - 1) / win, nwx = (gw + win - 1) / win;
    *rm = RowMap{(int)win, (int)gh, (int)gw, (int)nwx, (int)(nwx * nwy)};
    *Z = B * nwx * nwy;
    *N = win * win;
    return VAH_OK;
}
}  // namespace attn
}  // namespace vah

extern "C" {

static int attn_fwd_impl(const char *fn, const void *q, const void *k, const void *v, int64_t ld,
                         vah::attn::RowMap rm, int64_t B, int64_t H, int64_t N, float scale,
                         void *vt_ws, void *out, int64_t ld_out, float *lse, void *stream);

int vah_attn_fwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride,
                      int64_t B, int64_t H, int64_t N, float scale, void *vt_ws, void *out,
                      int64_t ld_out, float *lse, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_attn_fwd_bf16";
    if (N > 0 && batch_stride != N * ld) return fail(VAH_E_SHAPE, "%s: batch_stride must be N*ld", fn);
    return attn_fwd_impl(fn, q, k, v, ld, attn::RowMap{0, 0, 0, 0, 0}, B, H, N, scale, vt_ws, out, ld_out, lse, stream);
}

int vah_attn_bias_fwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride, int64_t B, int64_t H,
                           int64_t N, float scale, const void *bias, int64_t ldb, void *out, int64_t ld_out, float *lse,
                           void *stream) {
    using namespace vah;
    using namespace vah::attn;
    clear_error();
    const char *fn = "vah_attn_bias_fwd_bf16";
    if (N > 0 && batch_stride != N * ld) return fail(VAH_E_SHAPE, "%s: batch_stride must be N*ld", fn);
    if (B < 0 || H < 1 || N < 0 || ld < H * kHD || ld_out < H * kHD || B > 65535 || H > 65535 || N >= (1 << 24) || ldb < N ||
        ldb % 64)
        return fail(VAH_E_SHAPE, "%s: bad dims (ldb must be a multiple of 64 >= N)", fn);
    if (B == 0 || N == 0) return VAH_OK;
    if (!q || !k || !v || !bias || !out || !lse) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 || (ld % 8) || ((uintptr_t)out | (uintptr_t)bias) % 8 || (ld_out % 4))
        return fail(VAH_E_ALIGN, "%s: q/k/v need 16-byte aligned rows (ld %% 8 == 0), out / bias 8-byte", fn);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("attn_bias_fwd_bf16", 4 * B * H * N * kHD * 2 + B * H * N * 4 + B * H * N * N * 2, st, 0, 4 * B * H * N * N * kHD);
    return attn_fwd_seq(q, k, v, ld, B, H, N, scale, bias, ldb, out, ld_out, lse, st);
}

int vah_attn_win_fwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t B,
                          int64_t grid_h, int64_t grid_w, int64_t win, int64_t H, float scale,
                          void *vt_ws, void *out, int64_t ld_out, float *lse, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_attn_win_fwd_bf16";
    attn::RowMap rm;
    int64_t Z = 0, N = 0;
    if (win < 1) return fail(VAH_E_SHAPE, "%s: win must be >= 1", fn);
    if (int rc = attn::make_rowmap(fn, win, B, grid_h, grid_w, &Z, &N, &rm)) return rc;
    // windows of <= 224 tokens: one workgroup per (window, head) with K and V resident (attn_win.hip); vt_ws unused
    if (N <= 224 && Z >= 1 && Z <= 65535 && H >= 1 && H <= 65535 && ld >= H * attn::kHD && ld_out >= H * attn::kHD) {
        if (!q || !k || !v || !out || !lse) return fail(VAH_E_NULL, "%s: null pointer", fn);
        if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 || (ld % 8) || ((uintptr_t)out % 8) || (ld_out % 4))
            return fail(VAH_E_ALIGN, "%s: q/k/v need 16-byte aligned rows (ld %% 8 == 0), out 8-byte", fn);
        LaunchScope scope("attn_win_fwd_bf16", 4 * Z * H * N * attn::kHD * 2 + Z * H * N * 4, (hipStream_t)stream, 0,
                          4 * Z * H * N * N * attn::kHD);
        return attn::attn_win_fwd_resident(q, k, v, ld, rm, Z, H, N, scale, out, ld_out, lse, (hipStream_t)stream);
    }
    return attn_fwd_impl(fn, q, k, v, ld, rm, Z, H, N, scale, vt_ws, out, ld_out, lse, stream);
}

static int attn_fwd_impl(const char *fn, const void *q, const void *k, const void *v, int64_t ld,
                         vah::attn::RowMap rm, int64_t B, int64_t H, int64_t N, float scale,
                         void *vt_ws, void *out, int64_t ld_out, float *lse, void *stream) {
    using namespace vah;
    using namespace vah::attn;
    if (B < 0 || H < 1 || N < 0 || ld < H * kHD || ld_out < H * kHD || B > 65535 || H > 65535)
        return fail(VAH_E_SHAPE, "%s: bad dims B=%lld H=%lld N=%lld ld=%lld", fn, (long long)B,
                    (long long)H, (long long)N, (long long)ld);
    if (B == 0 || N == 0) return VAH_OK;
    if (!q || !k || !v || !vt_ws || !out || !lse) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)vt_ws) % 16 || (ld % 8) ||
        ((uintptr_t)out % 8) || (ld_out % 4))
        return fail(VAH_E_ALIGN, "%s: q/k/v need 16-byte aligned rows (ld %% 8 == 0), out 8-byte", fn);
    if (N >= (1 << 24)) return fail(VAH_E_SHAPE, "%s: N too large", fn);
    hipStream_t st = (hipStream_t)stream;
    if (rm.win == 0) {
        // whole sequences: the lean kernel of attn_flash.hip (no V^T workspace, no transpose launch)
        LaunchScope scope("attn_fwd_bf16", 4 * B * H * N * kHD * 2 + B * H * N * 4, st, 0, 4 * B * H * N * N * kHD);
        return attn_fwd_seq(q, k, v, ld, B, H, N, scale, nullptr, 0, out, ld_out, lse, st);
    }
    const int Np = (int)vah_attn_padded_len(N);
    {
        LaunchScope scope("attn_transpose_bf16", 2 * B * H * N * kHD * 2, st);
        hipLaunchKernelGGL(transpose_to_dn, dim3(Np / 64, (unsigned)H, (unsigned)B), dim3(256), 0, st,
                           (const __bf16 *)v, ld, rm, (int)N, Np, (int)H, (__bf16 *)vt_ws);
        if (int rc = check_launch(fn)) return rc;
    }
    const float scale_log2 = scale * 1.4426950408889634f;
    // algorithmic bytes: q, k, v read once, out written once (bf16) + lse
    // flops: QK^T + PV = 4 * B * heads * N^2 * 64; windows: N = win^2 with the padded keys
    LaunchScope scope(rm.win ? "attn_win_fwd_bf16" : "attn_fwd_bf16", 4 * B * H * N * kHD * 2 + B * H * N * 4, st, 0,
                      4 * B * H * N * N * kHD);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((N + kQBlock - 1) / kQBlock), (unsigned)H, (unsigned)B),
                       dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k, (const __bf16 *)vt_ws, ld,
                       rm, (int)N, Np, (int)H, scale_log2, (__bf16 *)out, ld_out, lse);
    return check_launch(fn);
}

}  // extern "C"
