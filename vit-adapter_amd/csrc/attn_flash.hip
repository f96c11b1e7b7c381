// Global (whole-sequence) attention forward and backward for head_dim 64, bf16 in / fp32 accumulate.
//
// Arithmetic of the reference's Attention (/root/reference/detection/mmdet_custom/models/backbones/base/vit.py:
// 83-88): softmax(q k^T * scale) v.  At head_dim 64 this operator is bound by the VALU, not the matrix pipe: a 32 x 32
// block of S costs 8 MFMAs (256 cycles) and 16 exp2 + 16 fma + 16 add + max + convert per lane (>= 400 cycles), so
// the loop below is written to the VALU instruction count (tools/isa_loop_mix.py), not to the MFMA count:
//   * one specialisation per addressing mode - no window arithmetic (integer divisions) in the key loop;
//   * K and V tiles row-major in LDS; V^T fragments by ds_read_b64_tr_b16 (no V^T copy, no transpose kernel);
//   * rows beyond N are CLAMPED to row N - 1 when staged (never zero-filled): their keys are masked to -inf in the
//     last tile, so P = 0 meets a finite V row;
//   * the cross-half maximum is one v_permlane32_swap (VALU) instead of a ds_bpermute round trip;
//   * the running output is rescaled only in iterations where some row's maximum moved (wave-uniform branch).
#include "attn_common.h"
#include "common.h"

namespace vah {
namespace attn {
namespace {

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;

__device__ __forceinline__ const __bf16 *tile_lane_base(const __bf16 *tile, int lane) {
    const int grp = lane >> 4, i16 = lane & 15;
    return tile + (4 * (grp >> 1) + (i16 >> 2)) * kPadRow + 16 * (grp & 1) + 4 * (i16 & 3);
}
// A operand T^T[m = d][k = token] of a row-major LDS tile T[token][d] in the k order of pack_half (attn_win.hip)
__device__ __forceinline__ bf16x8 load_tr(const __bf16 *base, int tok0, int db) {
    const __bf16 *p0 = base + tok0 * kPadRow + 32 * db;
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p0);
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p0 + 8 * kPadRow));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
}
// Exchange across the two 32-lane halves on the VALU: after v_permlane32_swap a = [x.lo, x.lo], b = [x.hi, x.hi].
// Inline asm: with the builtin and one value for both operands the compiler folds max(a, b) to a.  The s_nop covers
// the VALU-write -> permlane-read hazard the compiler would otherwise schedule around.
__device__ __forceinline__ void swap_halves(float x, float &a, float &b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float max_halves(float x) {
    float a, b;
    swap_halves(x, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float sum_halves(float x) {
    float a, b;
    swap_halves(x, a, b);
    return a + b;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kQB = 128, kKT = 64;

// BIAS: an additive term per (head, query, key) - BEiT's relative position bias (base/beit.py:120-144): `bias` holds
// bias * log2(e) as bf16, (H, N, ldb) with ldb a multiple of 64 >= N (columns beyond N unused)
template <bool BIAS>
__global__ __launch_bounds__(256) void attn_fwd_seq_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v, int64_t ld, int N, int H,
    float scale_log2, const __bf16 *__restrict__ bias, int64_t ldb, __bf16 *__restrict__ out, int64_t ld_out,
    float *__restrict__ lse) {
    __shared__ __attribute__((aligned(16))) __bf16 s_k2[2][kKT * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_v2[2][kKT * kPadRow];
    const int h = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int qrow = blockIdx.x * kQB + wave * 32 + r;
    const int64_t seq0 = (int64_t)b * N;
    const int64_t gq = seq0 + min(qrow, N - 1);

    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8 *>(q + gq * ld + h * kHD + 16 * kk + 8 * hf);

    // staging: 512 pieces of 16 bytes per tile and matrix, two per thread (rows sr and sr + 32)
    const int sr = threadIdx.x >> 3, sc = (threadIdx.x & 7) * 8;
    const __bf16 *kp = k + h * kHD + sc, *vp = v + h * kHD + sc;
    bf16x8 pk0, pk1, pv0, pv1;
    auto fetch = [&](int key0) {
        const int64_t g0 = (seq0 + min(key0 + sr, N - 1)) * ld, g1 = (seq0 + min(key0 + sr + 32, N - 1)) * ld;
        pk0 = *reinterpret_cast<const bf16x8 *>(kp + g0);
        pk1 = *reinterpret_cast<const bf16x8 *>(kp + g1);
        pv0 = *reinterpret_cast<const bf16x8 *>(vp + g0);
        pv1 = *reinterpret_cast<const bf16x8 *>(vp + g1);
    };
    auto commit = [&](int buf) {
        *reinterpret_cast<bf16x8 *>(s_k2[buf] + sr * kPadRow + sc) = pk0;
        *reinterpret_cast<bf16x8 *>(s_k2[buf] + (sr + 32) * kPadRow + sc) = pk1;
        *reinterpret_cast<bf16x8 *>(s_v2[buf] + sr * kPadRow + sc) = pv0;
        *reinterpret_cast<bf16x8 *>(s_v2[buf] + (sr + 32) * kPadRow + sc) = pv1;
    };

    f32x16 o[2] = {zero16(), zero16()};
    f32x16 lsum = zero16();
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.f;
    const f32x2 sc2 = {scale_log2, scale_log2};
    float m_run = -INFINITY;
    const int ntiles = (N + kKT - 1) / kKT;
    fetch(0);
    commit(0);
    if (ntiles > 1) fetch(kKT);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                 // tile t visible; every wave is done with tile t - 1
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);
            if (t + 2 < ntiles) fetch((t + 2) * kKT);
        }
        const __bf16 *s_k = s_k2[t & 1];
        const __bf16 *vbase = tile_lane_base(s_v2[t & 1], lane);
        bf16x4 bv[2][4];
        if constexpr (BIAS) {           // this lane's query row, 4 consecutive keys per register group
            const __bf16 *bp = bias + ((int64_t)h * N + min(qrow, N - 1)) * ldb + t * kKT + 4 * hf;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[kb][g] = *reinterpret_cast<const bf16x4 *>(bp + 32 * kb + 8 * g);
        }

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                s[kb] = mfma(*reinterpret_cast<const bf16x8 *>(s_k + (kb * 32 + r) * kPadRow + 16 * kk + 8 * hf), qf[kk], s[kb]);
        }
        if constexpr (BIAS) {           // scores in the log2 domain from here on
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 x = f32x2{s[kb][i], s[kb][i + 1]} * sc2 + f32x2{(float)bv[kb][i >> 2][i & 3], (float)bv[kb][i >> 2][(i & 3) + 1]};
                    s[kb][i] = x[0];
                    s[kb][i + 1] = x[1];
                }
        }
        if (t == ntiles - 1) {
            const int key0 = t * kKT;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + kb * 32 + crow(i, hf) >= N) s[kb][i] = -INFINITY;
        }
        // four independent chains (a single chain of 16 dependent v_max3 is latency, not issue, bound)
        float mq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mq[c] = fmaxf(fmaxf(s[0][4 * c], s[0][4 * c + 1]), fmaxf(s[0][4 * c + 2], s[0][4 * c + 3]));
            mq[c] = fmaxf(mq[c], fmaxf(fmaxf(s[1][4 * c], s[1][4 * c + 1]), fmaxf(s[1][4 * c + 2], s[1][4 * c + 3])));
        }
        const float mx = max_halves(fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3])));
        const float m_new = fmaxf(m_run, BIAS ? mx : mx * scale_log2);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run)) {                // some row's maximum moved: rescale
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);    // first tile: exp2(-inf) = 0
            lsum[0] *= alpha;                                             // every row of lsum is the same sum: row 0 is read
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            m_run = m_new;
        }
        // P^T = exp2(S^T scale - m): packed fma (two elements per VALU instruction), one v_exp each
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 x = {s[kb][i], s[kb][i + 1]};
                if constexpr (BIAS) x = x - f32x2{m_run, m_run};
                else x = x * sc2 - f32x2{m_run, m_run};
                s[kb][i] = __builtin_amdgcn_exp2f(x[0]);
                s[kb][i + 1] = __builtin_amdgcn_exp2f(x[1]);
            }
        // O^T += V^T P^T; the row sums of P ride along as a third block whose A operand is all ones (4 MFMAs on a pipe
        // with slack instead of 32 VALU adds on the pipe that bounds the loop), from the same bf16 P the product uses
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack_half(s[kb], sp);
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = mfma(load_tr(vbase, kb * 32 + 16 * sp, db), pf, o[db]);
                lsum = mfma(ones, pf, lsum);
            }
    }

    const float l_tot = lsum[0];      // the MFMA has summed over all keys: every row of column r is query r's sum
    const float inv = 1.f / l_tot;
    if (qrow < N) {
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
    }
}


// ---------------------------------------------------------------------------------------
// Backward, whole sequences.  Two kernels (dQ by query block, dK/dV by key block; S and dP are formed in both - the
// alternative, one pass with dQ accumulated across key blocks, needs 32 x 25 MB of fp32 atomics per call at this
// shape).  Against the general kernels of attn_bwd.hip: no transposed copies (K^T, Q^T, dO^T come from the row-major
// LDS tiles by ds_read_b64_tr_b16), so no prologue kernel - delta = rowsum(dO o O) is formed by the dQ kernel for its
// own queries and handed to the dK/dV kernel through `delta`; no window arithmetic in the loops; staged rows beyond N
// are clamped, not zero-filled.
// ---------------------------------------------------------------------------------------
// BIAS: scores carry `bias` as in the forward; dS (= d loss / d bias of this image) is written to ds_out
// (B, H, N, ldb) bf16, the caller sums it over the batch
template <bool BIAS>
__global__ __launch_bounds__(256) void attn_bwd_dq_seq_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v, int64_t ld,
    const __bf16 *__restrict__ o, const __bf16 *__restrict__ d_o, int64_t ld_out, const float *__restrict__ lse, int N, int H,
    float scale, float scale_log2, const __bf16 *__restrict__ bias, int64_t ldb, __bf16 *__restrict__ ds_out,
    float *__restrict__ delta, __bf16 *__restrict__ dq, int64_t ld_d) {
    __shared__ __attribute__((aligned(16))) __bf16 s_k2[2][kKT * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_v2[2][kKT * kPadRow];
    const int h = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int qrow = blockIdx.x * kQB + wave * 32 + r;
    const int64_t seq0 = (int64_t)b * N;
    const int64_t gq = seq0 + min(qrow, N - 1);

    bf16x8 qf[4], dof[4];
    float part = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *reinterpret_cast<const bf16x8 *>(q + gq * ld + h * kHD + 16 * kk + 8 * hf);
        dof[kk] = *reinterpret_cast<const bf16x8 *>(d_o + gq * ld_out + h * kHD + 16 * kk + 8 * hf);
        const bf16x8 of = *reinterpret_cast<const bf16x8 *>(o + gq * ld_out + h * kHD + 16 * kk + 8 * hf);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)of[j] * (float)dof[kk][j];
    }
    const float delta_q = sum_halves(part);
    const float lse_q = lse[((int64_t)b * H + h) * N + min(qrow, N - 1)];
    if (hf == 0 && qrow < N) delta[((int64_t)b * H + h) * N + qrow] = delta_q;

    const int sr = threadIdx.x >> 3, sc = (threadIdx.x & 7) * 8;
    const __bf16 *kp = k + h * kHD + sc, *vp = v + h * kHD + sc;
    bf16x8 pk0, pk1, pv0, pv1;
    auto fetch = [&](int key0) {
        const int64_t g0 = (seq0 + min(key0 + sr, N - 1)) * ld, g1 = (seq0 + min(key0 + sr + 32, N - 1)) * ld;
        pk0 = *reinterpret_cast<const bf16x8 *>(kp + g0);
        pk1 = *reinterpret_cast<const bf16x8 *>(kp + g1);
        pv0 = *reinterpret_cast<const bf16x8 *>(vp + g0);
        pv1 = *reinterpret_cast<const bf16x8 *>(vp + g1);
    };
    auto commit = [&](int buf) {
        *reinterpret_cast<bf16x8 *>(s_k2[buf] + sr * kPadRow + sc) = pk0;
        *reinterpret_cast<bf16x8 *>(s_k2[buf] + (sr + 32) * kPadRow + sc) = pk1;
        *reinterpret_cast<bf16x8 *>(s_v2[buf] + sr * kPadRow + sc) = pv0;
        *reinterpret_cast<bf16x8 *>(s_v2[buf] + (sr + 32) * kPadRow + sc) = pv1;
    };

    f32x16 acc[2] = {zero16(), zero16()};
    const f32x2 sc2 = {scale_log2, scale_log2}, ls2 = {lse_q, lse_q}, de2 = {delta_q, delta_q};
    const int ntiles = (N + kKT - 1) / kKT;
    fetch(0);
    commit(0);
    if (ntiles > 1) fetch(kKT);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);
            if (t + 2 < ntiles) fetch((t + 2) * kKT);
        }
        const __bf16 *s_k = s_k2[t & 1], *s_v = s_v2[t & 1];
        const __bf16 *kbase = tile_lane_base(s_k, lane);
        bf16x4 bv[2][4];
        const int64_t brow = ((int64_t)h * N + min(qrow, N - 1)) * ldb + t * kKT + 4 * hf;
        if constexpr (BIAS) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[kb][g] = *reinterpret_cast<const bf16x4 *>(bias + brow + 32 * kb + 8 * g);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = mfma(*reinterpret_cast<const bf16x8 *>(s_k + (kb * 32 + r) * kPadRow + 16 * kk + 8 * hf), qf[kk], s);
                dp = mfma(*reinterpret_cast<const bf16x8 *>(s_v + (kb * 32 + r) * kPadRow + 16 * kk + 8 * hf), dof[kk], dp);
            }
#pragma unroll
            for (int i = 0; i < 16; i += 2) {                           // dS^T = P^T o (dP^T - delta)
                f32x2 x = f32x2{s[i], s[i + 1]} * sc2 - ls2;
                if constexpr (BIAS) x = x + f32x2{(float)bv[kb][i >> 2][i & 3], (float)bv[kb][i >> 2][(i & 3) + 1]};
                const f32x2 pr = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                const f32x2 ds = pr * (f32x2{dp[i], dp[i + 1]} - de2);
                s[i] = ds[0];
                s[i + 1] = ds[1];
            }
            if (t == ntiles - 1) {                                      // clamped rows beyond N are real K rows: mask
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t * kKT + kb * 32 + crow(i, hf) >= N) s[i] = 0.f;
            }
            if constexpr (BIAS) {
                if (qrow < N) {
                    __bf16 *dp_ = ds_out + (int64_t)b * H * N * ldb + brow + 32 * kb;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[j] = (__bf16)s[4 * g + j];
                        *reinterpret_cast<bf16x4 *>(dp_ + 8 * g) = w;
                    }
                }
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack_half(s, sp);
#pragma unroll
                for (int db = 0; db < 2; ++db) acc[db] = mfma(load_tr(kbase, kb * 32 + 16 * sp, db), pf, acc[db]);
            }
        }
    }
    if (qrow < N) {
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

// dK, dV: workgroup = 128 keys (32 per wave), loop over query tiles of 64
// BIAS: bias_t = the forward's bias TRANSPOSED, (H, N keys, ldb queries): this lane's key row, 4 consecutive queries
template <bool BIAS>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_seq_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v, int64_t ld,
    const __bf16 *__restrict__ d_o, int64_t ld_out, const float *__restrict__ lse, const float *__restrict__ delta, int N, int H,
    float scale, float scale_log2, const __bf16 *__restrict__ bias_t, int64_t ldb, __bf16 *__restrict__ dk,
    __bf16 *__restrict__ dv, int64_t ld_d) {
    __shared__ __attribute__((aligned(16))) __bf16 s_q2[2][kKT * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_do2[2][kKT * kPadRow];
    __shared__ __attribute__((aligned(16))) float s_lse2[2][kKT];
    __shared__ __attribute__((aligned(16))) float s_delta2[2][kKT];
    const int h = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int krow = blockIdx.x * kQB + wave * 32 + r;
    const int64_t seq0 = (int64_t)b * N;
    const int64_t gk = seq0 + min(krow, N - 1);

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *reinterpret_cast<const bf16x8 *>(k + gk * ld + h * kHD + 16 * kk + 8 * hf);
        vf[kk] = *reinterpret_cast<const bf16x8 *>(v + gk * ld + h * kHD + 16 * kk + 8 * hf);
    }
    const int sr = threadIdx.x >> 3, sc = (threadIdx.x & 7) * 8;
    const __bf16 *qp = q + h * kHD + sc, *dop = d_o + h * kHD + sc;
    const float *lseb = lse + ((int64_t)b * H + h) * N, *delb = delta + ((int64_t)b * H + h) * N;
    bf16x8 pq0, pq1, pd0, pd1;
    float plse = 0.f, pdel = 0.f;
    auto fetch = [&](int q0) {
        const int64_t r0 = seq0 + min(q0 + sr, N - 1), r1 = seq0 + min(q0 + sr + 32, N - 1);
        pq0 = *reinterpret_cast<const bf16x8 *>(qp + r0 * ld);
        pq1 = *reinterpret_cast<const bf16x8 *>(qp + r1 * ld);
        pd0 = *reinterpret_cast<const bf16x8 *>(dop + r0 * ld_out);
        pd1 = *reinterpret_cast<const bf16x8 *>(dop + r1 * ld_out);
        if (threadIdx.x < kKT) {
            const int n = q0 + threadIdx.x;
            plse = n < N ? lseb[n] : INFINITY;                  // +inf: p = 0 for queries that do not exist
            pdel = n < N ? delb[n] : 0.f;
        }
    };
    auto commit = [&](int buf) {
        *reinterpret_cast<bf16x8 *>(s_q2[buf] + sr * kPadRow + sc) = pq0;
        *reinterpret_cast<bf16x8 *>(s_q2[buf] + (sr + 32) * kPadRow + sc) = pq1;
        *reinterpret_cast<bf16x8 *>(s_do2[buf] + sr * kPadRow + sc) = pd0;
        *reinterpret_cast<bf16x8 *>(s_do2[buf] + (sr + 32) * kPadRow + sc) = pd1;
        if (threadIdx.x < kKT) {
            s_lse2[buf][threadIdx.x] = plse;
            s_delta2[buf][threadIdx.x] = pdel;
        }
    };

    f32x16 dkt[2] = {zero16(), zero16()}, dvt[2] = {zero16(), zero16()};
    const f32x2 sc2 = {scale_log2, scale_log2};
    const int ntiles = (N + kKT - 1) / kKT;
    fetch(0);
    commit(0);
    if (ntiles > 1) fetch(kKT);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        if (t + 1 < ntiles) {
            commit((t + 1) & 1);
            if (t + 2 < ntiles) fetch((t + 2) * kKT);
        }
        const __bf16 *s_q = s_q2[t & 1], *s_do = s_do2[t & 1];
        const float *s_lse = s_lse2[t & 1], *s_delta = s_delta2[t & 1];
        const __bf16 *qbase = tile_lane_base(s_q, lane), *dobase = tile_lane_base(s_do, lane);
        bf16x4 bv[2][4];
        if constexpr (BIAS) {
            const __bf16 *bp = bias_t + ((int64_t)h * N + min(krow, N - 1)) * ldb + t * kKT + 4 * hf;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[qb][g] = *reinterpret_cast<const bf16x4 *>(bp + 32 * qb + 8 * g);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = mfma(*reinterpret_cast<const bf16x8 *>(s_q + (qb * 32 + r) * kPadRow + 16 * kk + 8 * hf), kf[kk], s);      // S[q][key]
                dp = mfma(*reinterpret_cast<const bf16x8 *>(s_do + (qb * 32 + r) * kPadRow + 16 * kk + 8 * hf), vf[kk], dp);   // dP[q][key]
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l4 = *reinterpret_cast<const float4 *>(s_lse + qb * 32 + 8 * g + 4 * hf);
                const float4 d4 = *reinterpret_cast<const float4 *>(s_delta + qb * 32 + 8 * g + 4 * hf);
                const f32x2 l01 = {l4.x, l4.y}, l23 = {l4.z, l4.w}, d01 = {d4.x, d4.y}, d23 = {d4.z, d4.w};
                f32x2 x0 = f32x2{s[4 * g], s[4 * g + 1]} * sc2 - l01, x1 = f32x2{s[4 * g + 2], s[4 * g + 3]} * sc2 - l23;
                if constexpr (BIAS) {
                    x0 = x0 + f32x2{(float)bv[qb][g][0], (float)bv[qb][g][1]};
                    x1 = x1 + f32x2{(float)bv[qb][g][2], (float)bv[qb][g][3]};
                }
                const f32x2 p0 = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
                const f32x2 p1 = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
                const f32x2 e0 = p0 * (f32x2{dp[4 * g], dp[4 * g + 1]} - d01), e1 = p1 * (f32x2{dp[4 * g + 2], dp[4 * g + 3]} - d23);
                s[4 * g] = p0[0], s[4 * g + 1] = p0[1], s[4 * g + 2] = p1[0], s[4 * g + 3] = p1[1];            // P
                dp[4 * g] = e0[0], dp[4 * g + 1] = e0[1], dp[4 * g + 2] = e1[0], dp[4 * g + 3] = e1[1];        // dS
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack_half(s, sp), dsf = pack_half(dp, sp);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dvt[db] = mfma(load_tr(dobase, qb * 32 + 16 * sp, db), pf, dvt[db]);      // dV^T[d][key] += dO^T P
                    dkt[db] = mfma(load_tr(qbase, qb * 32 + 16 * sp, db), dsf, dkt[db]);      // dK^T[d][key] += Q^T dS
                }
            }
        }
    }
    if (krow < N) {
        __bf16 *pk = dk + gk * ld_d + (int64_t)h * kHD, *pv = dv + gk * ld_d + (int64_t)h * kHD;
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

int attn_fwd_seq(const void *q, const void *k, const void *v, int64_t ld, int64_t B, int64_t H, int64_t N, float scale,
                 const void *bias, int64_t ldb, void *out, int64_t ld_out, float *lse, hipStream_t st) {
    const dim3 grid((unsigned)((N + kQB - 1) / kQB), (unsigned)H, (unsigned)B);
    if (bias)
        hipLaunchKernelGGL(attn_fwd_seq_kernel<true>, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                           (const __bf16 *)v, ld, (int)N, (int)H, scale * 1.4426950408889634f, (const __bf16 *)bias, ldb,
                           (__bf16 *)out, ld_out, lse);
    else
        hipLaunchKernelGGL(attn_fwd_seq_kernel<false>, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                           (const __bf16 *)v, ld, (int)N, (int)H, scale * 1.4426950408889634f, (const __bf16 *)nullptr,
                           (int64_t)0, (__bf16 *)out, ld_out, lse);
    return check_launch("attn_fwd_seq");
}

int attn_bwd_seq(const void *q, const void *k, const void *v, int64_t ld, const void *o, const void *d_o, int64_t ld_out,
                 const float *lse, int64_t B, int64_t H, int64_t N, float scale, const void *bias, const void *bias_t,
                 int64_t ldb, void *ds_out, float *delta, void *dq, void *dk, void *dv, int64_t ld_d, hipStream_t st) {
    const dim3 grid((unsigned)((N + kQB - 1) / kQB), (unsigned)H, (unsigned)B);
    const float scale_log2 = scale * 1.4426950408889634f;
    {
        // useful flops of the whole backward = 2.5x the forward (S, dP, dV, dK, dQ products); both kernels form S and
        // dP: dq runs 3 products (6 B H N^2 64), dkdv 4 (8 B H N^2 64)
        LaunchScope scope("attn_bwd_dq_bf16", 6 * B * H * N * kHD * 2, st, 0, 6 * B * H * N * N * kHD);
        if (bias)
            hipLaunchKernelGGL(attn_bwd_dq_seq_kernel<true>, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                               (const __bf16 *)v, ld, (const __bf16 *)o, (const __bf16 *)d_o, ld_out, lse, (int)N, (int)H, scale,
                               scale_log2, (const __bf16 *)bias, ldb, (__bf16 *)ds_out, delta, (__bf16 *)dq, ld_d);
        else
            hipLaunchKernelGGL(attn_bwd_dq_seq_kernel<false>, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                               (const __bf16 *)v, ld, (const __bf16 *)o, (const __bf16 *)d_o, ld_out, lse, (int)N, (int)H, scale,
                               scale_log2, (const __bf16 *)nullptr, (int64_t)0, (__bf16 *)nullptr, delta, (__bf16 *)dq, ld_d);
        if (int rc = check_launch("attn_bwd_dq_seq")) return rc;
    }
    LaunchScope scope("attn_bwd_dkdv_bf16", 8 * B * H * N * kHD * 2, st, 0, 8 * B * H * N * N * kHD);
    if (bias)
        hipLaunchKernelGGL(attn_bwd_dkdv_seq_kernel<true>, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                           (const __bf16 *)v, ld, (const __bf16 *)d_o, ld_out, lse, delta, (int)N, (int)H, scale, scale_log2,
                           (const __bf16 *)bias_t, ldb, (__bf16 *)dk, (__bf16 *)dv, ld_d);
    else
        hipLaunchKernelGGL(attn_bwd_dkdv_seq_kernel<false>, grid, dim3(256), 0, st, (const __bf16 *)q, (const __bf16 *)k,
                           (const __bf16 *)v, ld, (const __bf16 *)d_o, ld_out, lse, delta, (int)N, (int)H, scale, scale_log2,
                           (const __bf16 *)nullptr, (int64_t)0, (__bf16 *)dk, (__bf16 *)dv, ld_d);
    return check_launch("attn_bwd_dkdv_seq");
}

}  // namespace attn
}  // namespace vah
