// Fused softmax attention backward for gfx950 (bf16, head_dim 64): gradients of
//   out = softmax(q k^T * scale) v
// w.r.t. q, k, v, written straight into a packed (B, N, 3, heads, 64) gradient buffer.
//
// The probabilities are recomputed from q, k and the forward's log-sum-exp (no N x N tensor is
// stored):  P = exp2(c q.k - LSE),  dP = dO V^T,  dS = P (dP - delta),  delta = rowsum(dO o O),
//   dV = P^T dO,   dK = scale dS^T Q,   dQ = scale dS K.
// Two kernels, no atomics, bitwise reproducible:
//   * attn_bwd_dq   : one workgroup per 128 QUERIES walks the keys (scores transposed, keys on
//                     registers / query on the lane, exactly like the forward), so LSE and delta
//                     are per-lane scalars and dS^T feeds  dQ^T = K^T dS^T  from registers.
//   * attn_bwd_dkdv : one workgroup per 128 KEYS walks the queries (scores NOT transposed:
//                     query on registers / key on the lane), so P and dS feed
//                     dV^T = dO^T P  and  dK^T = Q^T dS  from registers and dK / dV need no
//                     reduction across workgroups.
// The transposed operands (K^T, Q^T, dO^T as (B, heads, 64, Np)) are produced once per call by the
// small transpose kernel, so every LDS tile is filled with plain 16-byte copies.
#include "attn_common.h"
#include "common.h"

namespace vah {
namespace attn {
namespace {

// Backward prologue in ONE launch: K^T, Q^T, dO^T (64-token tiles) and, from the dO tile it already
// holds, delta = rowsum(dO * O).  Four small launches before (3 transposes + delta): on the 196-token
// windows they cost as much as the dQ kernel itself.
__global__ __launch_bounds__(256) void attn_bwd_prologue_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, int64_t ld, const __bf16 *__restrict__ o,
    const __bf16 *__restrict__ d_o, int64_t ld_out, RowMap rm, int N, int Np, int H, __bf16 *__restrict__ kt,
    __bf16 *__restrict__ qt, __bf16 *__restrict__ dot, float *__restrict__ delta) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[64 * kPadRow];
    const int ntile = Np / 64;
    const int which = blockIdx.x / ntile, n0 = (blockIdx.x - which * ntile) * 64;
    const int h = blockIdx.y, b = blockIdx.z;
    if (which == 0) {
        transpose_tile_to_dn(k, ld, rm, N, Np, H, kt, n0, h, b, tile);
    } else if (which == 1) {
        transpose_tile_to_dn(q, ld, rm, N, Np, H, qt, n0, h, b, tile);
    } else {
        transpose_tile_to_dn(d_o, ld_out, rm, N, Np, H, dot, n0, h, b, tile);
        // delta of the tile's 64 tokens: 4 threads per token, 16 channels each, dO from the LDS tile
        const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
        const int n = n0 + row;
        float acc = 0.f;
        const int64_t gr = n < N ? grow(rm, b, n, N) : -1;
        if (gr >= 0) {
            const __bf16 *po = o + gr * ld_out + (int64_t)h * kHD + part * 16;
            const __bf16 *pd = tile + row * kPadRow + part * 16;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(po + 8 * c);
                const bf16x8 g = *reinterpret_cast<const bf16x8 *>(pd + 8 * c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)g[j];
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (part == 0 && n < N) delta[((int64_t)b * H + h) * N + n] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// dQ: workgroup = 128 queries, loop over key tiles of 64
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    const __bf16 *__restrict__ kt, const __bf16 *__restrict__ d_o, int64_t ld, RowMap rm,
    int64_t ld_out, const float *__restrict__ lse, const float *__restrict__ delta, int N, int Np,
    int H, float scale, float scale_log2, __bf16 *__restrict__ dq, int64_t ld_d) {
    // double buffered (one barrier per key tile)
    __shared__ __attribute__((aligned(16))) __bf16 s_k2[2][64 * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_v2[2][64 * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_kt2[2][kHD * kPadT];

    const int h = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int qrow = blockIdx.x * 128 + wave * 32 + r;
    const int qload = min(qrow, N - 1);

    const __bf16 *qb = q + (int64_t)h * kHD;
    const __bf16 *kb = k + (int64_t)h * kHD;
    const __bf16 *vb = v + (int64_t)h * kHD;
    const __bf16 *ktb = kt + ((int64_t)(b * H + h) * kHD) * Np;
    const __bf16 *dob = d_o + (int64_t)h * kHD;
    const int64_t gq = grow(rm, b, qload, N);

    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[kk][j] = dof[kk][j] = (__bf16)0.f;
        if (gq >= 0) {
            qf[kk] = *reinterpret_cast<const bf16x8 *>(qb + gq * ld + 16 * kk + 8 * hf);
            dof[kk] = *reinterpret_cast<const bf16x8 *>(dob + gq * ld_out + 16 * kk + 8 * hf);
        }
    }
    const float lse_q = lse[((int64_t)b * H + h) * N + qload];
    const float delta_q = delta[((int64_t)b * H + h) * N + qload];
    f32x16 acc[2] = {zero16(), zero16()};

    const int c0 = threadIdx.x, c1 = threadIdx.x + 256;
    const int r0 = c0 >> 3, x0 = (c0 & 7) * 8, r1 = c1 >> 3, x1 = (c1 & 7) * 8;
    bf16x8 pk0, pk1, pv0, pv1, pt0, pt1;
    auto fetch = [&](int key0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pk0[j] = pk1[j] = pv0[j] = pv1[j] = (__bf16)0.f;
        const int64_t g0 = key0 + r0 < N ? grow(rm, b, key0 + r0, N) : -1;
        const int64_t g1 = key0 + r1 < N ? grow(rm, b, key0 + r1, N) : -1;
        if (g0 >= 0) {
            pk0 = *reinterpret_cast<const bf16x8 *>(kb + g0 * ld + x0);
            pv0 = *reinterpret_cast<const bf16x8 *>(vb + g0 * ld + x0);
        }
        if (g1 >= 0) {
            pk1 = *reinterpret_cast<const bf16x8 *>(kb + g1 * ld + x1);
            pv1 = *reinterpret_cast<const bf16x8 *>(vb + g1 * ld + x1);
        }
        pt0 = *reinterpret_cast<const bf16x8 *>(ktb + (int64_t)r0 * Np + key0 + x0);
        pt1 = *reinterpret_cast<const bf16x8 *>(ktb + (int64_t)r1 * Np + key0 + x1);
    };
    auto commit = [&](int buf) {
        __bf16 *s_k = s_k2[buf], *s_v = s_v2[buf], *s_kt = s_kt2[buf];
        *reinterpret_cast<bf16x8 *>(s_k + r0 * kPadRow + x0) = pk0;
        *reinterpret_cast<bf16x8 *>(s_k + r1 * kPadRow + x1) = pk1;
        *reinterpret_cast<bf16x8 *>(s_v + r0 * kPadRow + x0) = pv0;
        *reinterpret_cast<bf16x8 *>(s_v + r1 * kPadRow + x1) = pv1;
        const bf16x4 *a0 = reinterpret_cast<const bf16x4 *>(&pt0), *a1 = reinterpret_cast<const bf16x4 *>(&pt1);
        *reinterpret_cast<bf16x4 *>(s_kt + r0 * kPadT + x0) = a0[0];
        *reinterpret_cast<bf16x4 *>(s_kt + r0 * kPadT + x0 + 4) = a0[1];
        *reinterpret_cast<bf16x4 *>(s_kt + r1 * kPadT + x1) = a1[0];
        *reinterpret_cast<bf16x4 *>(s_kt + r1 * kPadT + x1 + 4) = a1[1];
    };

    const int ntiles = (N + 63) / 64;
    fetch(0);
    commit(0);
    if (ntiles > 1) fetch(64);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);
            if (t + 2 < ntiles) fetch((t + 2) * 64);
        }
        const __bf16 *s_k = s_k2[t & 1], *s_v = s_v2[t & 1], *s_kt = s_kt2[t & 1];
        const int key0 = t * 64;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 ak = *reinterpret_cast<const bf16x8 *>(s_k + (kbk * 32 + r) * kPadRow + 16 * kk + 8 * hf);
                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(s_v + (kbk * 32 + r) * kPadRow + 16 * kk + 8 * hf);
                s = mfma(ak, qf[kk], s);
                dp = mfma(av, dof[kk], dp);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool live = key0 + kbk * 32 + crow(i, hf) < N;
                const float p = live ? __builtin_amdgcn_exp2f(s[i] * scale_log2 - lse_q) : 0.f;
                s[i] = p * (dp[i] - delta_q);                       // dS^T
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack_half(s, sp);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 a = load_kperm(s_kt + (db * 32 + r) * kPadT + kbk * 32, sp, hf);
                    acc[db] = mfma(a, pf, acc[db]);
                }
            }
        }
    }
    if (qrow < N && gq >= 0) {
        __bf16 *op = dq + gq * ld_d + (int64_t)h * kHD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (__bf16)(acc[db][4 * g + j] * scale);
                *reinterpret_cast<bf16x4 *>(op + db * 32 + 8 * g + 4 * hf) = w;
            }
    }
}

// ---------------------------------------------------------------------------------------
// dK, dV: workgroup = 128 keys, loop over query tiles of 32
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v,
    const __bf16 *__restrict__ qt, const __bf16 *__restrict__ d_o, const __bf16 *__restrict__ dot,
    int64_t ld, RowMap rm, int64_t ld_out, const float *__restrict__ lse,
    const float *__restrict__ delta, int N, int Np, int H, float scale, float scale_log2,
    __bf16 *__restrict__ dk, __bf16 *__restrict__ dv, int64_t ld_d) {
    // double buffered (one barrier per query tile)
    __shared__ __attribute__((aligned(16))) __bf16 s_q2[2][32 * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_do2[2][32 * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_qt2[2][kHD * kPadT32];
    __shared__ __attribute__((aligned(16))) __bf16 s_dot2[2][kHD * kPadT32];
    __shared__ __attribute__((aligned(16))) float s_lse2[2][32];
    __shared__ __attribute__((aligned(16))) float s_delta2[2][32];

    const int h = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int krow = blockIdx.x * 128 + wave * 32 + r;              // this lane's key
    const int kload = min(krow, N - 1);
    const int64_t gk = grow(rm, b, kload, N);
    // a padded key is NOT masked: it takes part in the softmax with k = v = 0 (reference quirk),
    // only its gradient has nowhere to go
    const bool key_live = krow < N;

    const __bf16 *qb = q + (int64_t)h * kHD;
    const __bf16 *kb = k + (int64_t)h * kHD;
    const __bf16 *vb = v + (int64_t)h * kHD;
    const __bf16 *qtb = qt + ((int64_t)(b * H + h) * kHD) * Np;
    const __bf16 *dotb = dot + ((int64_t)(b * H + h) * kHD) * Np;
    const __bf16 *dob = d_o + (int64_t)h * kHD;
    const float *lseb = lse + ((int64_t)b * H + h) * N;
    const float *delb = delta + ((int64_t)b * H + h) * N;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[kk][j] = vf[kk][j] = (__bf16)0.f;
        if (gk >= 0) {
            kf[kk] = *reinterpret_cast<const bf16x8 *>(kb + gk * ld + 16 * kk + 8 * hf);
            vf[kk] = *reinterpret_cast<const bf16x8 *>(vb + gk * ld + 16 * kk + 8 * hf);
        }
    }
    f32x16 dkt[2] = {zero16(), zero16()}, dvt[2] = {zero16(), zero16()};

    // staging per query tile: Q, dO: 32 rows x 8 chunks = 256 chunks (1 per thread);
    // Q^T, dO^T: 64 d-rows x 4 chunks of 8 queries = 256 chunks (1 per thread)
    const int rr = threadIdx.x >> 3, rx = (threadIdx.x & 7) * 8;        // row-major tiles
    const int tr = threadIdx.x >> 2, tx = (threadIdx.x & 3) * 8;        // transposed tiles
    bf16x8 pq, pdo, pqt, pdot;
    float plse = INFINITY, pdel = 0.f;
    auto fetch = [&](int q0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pq[j] = pdo[j] = (__bf16)0.f;
        const int64_t gr = q0 + rr < N ? grow(rm, b, q0 + rr, N) : -1;
        if (gr >= 0) {
            pq = *reinterpret_cast<const bf16x8 *>(qb + gr * ld + rx);
            pdo = *reinterpret_cast<const bf16x8 *>(dob + gr * ld_out + rx);
        }
        pqt = *reinterpret_cast<const bf16x8 *>(qtb + (int64_t)tr * Np + q0 + tx);
        pdot = *reinterpret_cast<const bf16x8 *>(dotb + (int64_t)tr * Np + q0 + tx);
        if (threadIdx.x < 32) {
            const bool ok = q0 + (int)threadIdx.x < N;
            plse = ok ? lseb[q0 + threadIdx.x] : INFINITY;         // +inf -> p = 0 for padded queries
            pdel = ok ? delb[q0 + threadIdx.x] : 0.f;
        }
    };
    auto commit = [&](int buf) {
        __bf16 *s_q = s_q2[buf], *s_do = s_do2[buf], *s_qt = s_qt2[buf], *s_dot = s_dot2[buf];
        float *s_lse = s_lse2[buf], *s_delta = s_delta2[buf];
        *reinterpret_cast<bf16x8 *>(s_q + rr * kPadRow + rx) = pq;
        *reinterpret_cast<bf16x8 *>(s_do + rr * kPadRow + rx) = pdo;
        const bf16x4 *a = reinterpret_cast<const bf16x4 *>(&pqt), *g = reinterpret_cast<const bf16x4 *>(&pdot);
        *reinterpret_cast<bf16x4 *>(s_qt + tr * kPadT32 + tx) = a[0];
        *reinterpret_cast<bf16x4 *>(s_qt + tr * kPadT32 + tx + 4) = a[1];
        *reinterpret_cast<bf16x4 *>(s_dot + tr * kPadT32 + tx) = g[0];
        *reinterpret_cast<bf16x4 *>(s_dot + tr * kPadT32 + tx + 4) = g[1];
        if (threadIdx.x < 32) {
            s_lse[threadIdx.x] = plse;
            s_delta[threadIdx.x] = pdel;
        }
    };

    const int ntiles = (N + 31) / 32;
    fetch(0);
    commit(0);
    if (ntiles > 1) fetch(32);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);
            if (t + 2 < ntiles) fetch((t + 2) * 32);
        }
        const __bf16 *s_q = s_q2[t & 1], *s_do = s_do2[t & 1], *s_qt = s_qt2[t & 1], *s_dot = s_dot2[t & 1];
        const float *s_lse = s_lse2[t & 1], *s_delta = s_delta2[t & 1];

        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 aq = *reinterpret_cast<const bf16x8 *>(s_q + r * kPadRow + 16 * kk + 8 * hf);
            const bf16x8 ad = *reinterpret_cast<const bf16x8 *>(s_do + r * kPadRow + 16 * kk + 8 * hf);
            s = mfma(aq, kf[kk], s);          // S[query (reg)][key (lane)]
            dp = mfma(ad, vf[kk], dp);        // dP[query][key]
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4 *>(s_lse + 8 * g + 4 * hf);
            const float4 d4 = *reinterpret_cast<const float4 *>(s_delta + 8 * g + 4 * hf);
            const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * g + j;
                const float p = key_live ? __builtin_amdgcn_exp2f(s[i] * scale_log2 - ls[j]) : 0.f;
                s[i] = p;                                  // P
                dp[i] = p * (dp[i] - de[j]);               // dS
            }
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack_half(s, sp), dsf = pack_half(dp, sp);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bf16x8 ado = load_kperm(s_dot + (db * 32 + r) * kPadT32, sp, hf);
                const bf16x8 aq = load_kperm(s_qt + (db * 32 + r) * kPadT32, sp, hf);
                dvt[db] = mfma(ado, pf, dvt[db]);          // dV^T[d][key] += dO^T P
                dkt[db] = mfma(aq, dsf, dkt[db]);          // dK^T[d][key] += Q^T dS
            }
        }
    }
    if (key_live && gk >= 0) {
        __bf16 *pk = dk + gk * ld_d + (int64_t)h * kHD;
        __bf16 *pv = dv + gk * ld_d + (int64_t)h * kHD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wk[j] = (__bf16)(dkt[db][4 * g + j] * scale);
                    wv[j] = (__bf16)dvt[db][4 * g + j];
                }
                *reinterpret_cast<bf16x4 *>(pk + db * 32 + 8 * g + 4 * hf) = wk;
                *reinterpret_cast<bf16x4 *>(pv + db * 32 + 8 * g + 4 * hf) = wv;
            }
    }
}

}  // namespace
}  // namespace attn
}  // namespace vah

extern "C" {

int64_t vah_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t N) {
    const int64_t Np = (N + 63) / 64 * 64;
    return 3 * B * H * 64 * Np * 2 + (B * H * N * 4 + 15) / 16 * 16;
}

static int attn_bwd_impl(const char *fn, const void *q, const void *k, const void *v, int64_t ld,
                         vah::attn::RowMap rm, const void *out, const void *dout, int64_t ld_out,
                         const float *lse, int64_t B, int64_t H, int64_t N, float scale, void *ws, void *dq,
                         void *dk, void *dv, int64_t ld_d, void *stream) {
    using namespace vah;
    using namespace vah::attn;
    if (B < 0 || H < 1 || N < 0 || ld < H * kHD || ld_out < H * kHD || ld_d < H * kHD || B > 65535 || H > 65535)
        return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (B == 0 || N == 0) return VAH_OK;
    if (!q || !k || !v || !out || !dout || !lse || !ws || !dq || !dk || !dv)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)ws) % 16 ||
        (ld % 8) || (ld_out % 8) || (ld_d % 4) || ((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned operand", fn);
    if (N >= (1 << 24)) return fail(VAH_E_SHAPE, "%s: N too large", fn);
    hipStream_t st = (hipStream_t)stream;
    // whole sequences: the lean kernels of attn_flash.hip (no transposed copies; of ws only B*H*N floats for delta)
    if (rm.win == 0)
        return attn_bwd_seq(q, k, v, ld, out, dout, ld_out, lse, B, H, N, scale, nullptr, nullptr, 0, nullptr, (float *)ws, dq, dk, dv,
                            ld_d, st);
    const int Np = (int)((N + 63) / 64 * 64);
    const int64_t tsz = B * H * 64 * (int64_t)Np;
    __bf16 *kt = (__bf16 *)ws, *qt = kt + tsz, *dot = qt + tsz;
    float *delta = (float *)(dot + tsz);
    const dim3 tg(Np / 64, (unsigned)H, (unsigned)B);
    {
        LaunchScope scope("attn_transpose_bf16", 3 * 2 * B * H * N * kHD * 2, st);
        hipLaunchKernelGGL(attn_bwd_prologue_kernel, dim3(3 * tg.x, tg.y, tg.z), dim3(256), 0, st, (const __bf16 *)q,
                           (const __bf16 *)k, ld, (const __bf16 *)out, (const __bf16 *)dout, ld_out, rm, (int)N, Np,
                           (int)H, kt, qt, dot, delta);
        if (int rc = check_launch(fn)) return rc;
    }
    const float scale_log2 = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)H, (unsigned)B);
    {
        // useful flops of the whole backward = 2.5x the forward (S, dP, dV, dK, dQ products); both kernels
        // recompute S and dP: dq runs 3 products (6 B H N^2 64), dkdv 4 (8 B H N^2 64)
        LaunchScope scope(rm.win ? "attn_win_bwd_dq_bf16" : "attn_bwd_dq_bf16", 6 * B * H * N * kHD * 2, st, 0,
                          6 * B * H * N * N * kHD);
        hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                           (const __bf16 *)v, kt, (const __bf16 *)dout, ld, rm, ld_out, lse, delta,
                           (int)N, Np, (int)H, scale, scale_log2, (__bf16 *)dq, ld_d);
        if (int rc = check_launch(fn)) return rc;
    }
    LaunchScope scope(rm.win ? "attn_win_bwd_dkdv_bf16" : "attn_bwd_dkdv_bf16", 8 * B * H * N * kHD * 2, st, 0,
                      8 * B * H * N * N * kHD);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                       (const __bf16 *)v, qt, (const __bf16 *)dout, dot, ld, rm, ld_out, lse, delta,
                       (int)N, Np, (int)H, scale, scale_log2, (__bf16 *)dk, (__bf16 *)dv, ld_d);
    return check_launch(fn);
}

int vah_attn_bwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride,
                      const void *out, const void *dout, int64_t ld_out, const float *lse, int64_t B,
                      int64_t H, int64_t N, float scale, void *ws, void *dq, void *dk, void *dv,
                      int64_t ld_d, int64_t batch_stride_d, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_attn_bwd_bf16";
    if (N > 0 && (batch_stride != N * ld || batch_stride_d != N * ld_d))
        return fail(VAH_E_SHAPE, "%s: batch strides must be N*ld", fn);
    return attn_bwd_impl(fn, q, k, v, ld, attn::RowMap{0, 0, 0, 0, 0}, out, dout, ld_out, lse, B, H, N, scale, ws,
                         dq, dk, dv, ld_d, stream);
}

int vah_attn_bias_bwd_bf16(const void *q, const void *k, const void *v, int64_t ld, int64_t batch_stride, const void *out,
                           const void *dout, int64_t ld_out, const float *lse, int64_t B, int64_t H, int64_t N, float scale,
                           const void *bias, const void *bias_t, int64_t ldb, void *ds_out, float *delta_ws, void *dq, void *dk,
                           void *dv, int64_t ld_d, int64_t batch_stride_d, void *stream) {
    using namespace vah;
    using namespace vah::attn;
    clear_error();
    const char *fn = "vah_attn_bias_bwd_bf16";
    if (N > 0 && (batch_stride != N * ld || batch_stride_d != N * ld_d)) return fail(VAH_E_SHAPE, "%s: batch strides must be N*ld", fn);
    if (B < 0 || H < 1 || N < 0 || ld < H * kHD || ld_out < H * kHD || ld_d < H * kHD || B > 65535 || H > 65535 || N >= (1 << 24) ||
        ldb < N || ldb % 64)
        return fail(VAH_E_SHAPE, "%s: bad dims (ldb must be a multiple of 64 >= N)", fn);
    if (B == 0 || N == 0) return VAH_OK;
    if (!q || !k || !v || !out || !dout || !lse || !bias || !bias_t || !ds_out || !delta_ws || !dq || !dk || !dv)
        return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout) % 16 || (ld % 8) || (ld_out % 8) || (ld_d % 4) ||
        ((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)bias | (uintptr_t)bias_t | (uintptr_t)ds_out) % 8)
        return fail(VAH_E_ALIGN, "%s: misaligned operand", fn);
    return attn_bwd_seq(q, k, v, ld, out, dout, ld_out, lse, B, H, N, scale, bias, bias_t, ldb, ds_out, delta_ws, dq, dk, dv, ld_d,
                        (hipStream_t)stream);
}

int vah_attn_win_bwd_bf16(const void *q, const void *k, const void *v, int64_t ld, const void *out,
                          const void *dout, int64_t ld_out, const float *lse, int64_t B, int64_t grid_h,
                          int64_t grid_w, int64_t win, int64_t H, float scale, void *ws, void *dq,
                          void *dk, void *dv, int64_t ld_d, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_attn_win_bwd_bf16";
    attn::RowMap rm;
    int64_t Z = 0, N = 0;
    if (win < 1) return fail(VAH_E_SHAPE, "%s: win must be >= 1", fn);
    if (int rc = attn::make_rowmap(fn, win, B, grid_h, grid_w, &Z, &N, &rm)) return rc;
    // windows of <= 224 tokens: one kernel, one workgroup per (window, head), everything resident (attn_win.hip);
    // ws unused
    if (N <= 224 && Z >= 1 && Z <= 65535 && H >= 1 && H <= 65535 && ld >= H * attn::kHD && ld_out >= H * attn::kHD &&
        ld_d >= H * attn::kHD) {
        if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv) return fail(VAH_E_NULL, "%s: null pointer", fn);
        if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout) % 16 || (ld % 8) || (ld_out % 8) ||
            ((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 8 || (ld_d % 4))
            return fail(VAH_E_ALIGN, "%s: q/k/v/out/dout need 16-byte aligned rows, dq/dk/dv 8-byte", fn);
        // one kernel does the work of the prologue, dq and dkdv kernels: bytes = q, k, v, out, dout in, dq, dk, dv out;
        // flops (algorithmic) = S, dP, dQ, dK, dV = 10 N^2 64 per (window, head); the kernel forms S and dP twice
        LaunchScope scope("attn_win_bwd_bf16", 8 * Z * H * N * attn::kHD * 2 + Z * H * N * 4, (hipStream_t)stream, 0,
                          10 * Z * H * N * N * attn::kHD);
        return attn::attn_win_bwd_resident(q, k, v, ld, out, dout, ld_out, rm, lse, Z, H, N, scale, dq, dk, dv, ld_d,
                                           (hipStream_t)stream);
    }
    return attn_bwd_impl(fn, q, k, v, ld, rm, out, dout, ld_out, lse, Z, H, N, scale, ws, dq, dk, dv, ld_d, stream);
}

}  // extern "C"
