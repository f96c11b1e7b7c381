// Windowed attention forward with the whole window resident: one workgroup per (window, head).
//
// Arithmetic of the reference's WindowedAttention (/root/reference/detection/mmdet_custom/models/backbones/
// base/vit.py:136-167): softmax(q k^T * scale) v inside win x win windows of the token grid, the grid zero-padded
// AFTER the projection (padded tokens are real keys with k = v = 0: RowMap, attn_common.h).
//
// The general kernel (attn_fwd.hip) walks the keys in tiles of 64 with an online softmax, 128 queries per
// workgroup, V^T from a separate transpose kernel: for a 14 x 14 window (196 tokens) that is 2 query blocks x 4 key
// tiles - the last tile holds 4 keys, the second query block 68 queries: 59 % of the matrix work is padding - and
// five latency-bound launches per layer with the backward's.  Here:
//   * K and V of the window (<= 224 x 64 bf16 each) are staged ONCE, row-major, 144-byte rows;
//   * NB = ceil(N / 32) waves, wave w owns queries 32w .. 32w + 31 and walks ALL keys twice: row maxima of
//     S^T = K Q^T first, then S^T again with exp2 and the product (no running maximum, no rescale; S is not held - 7
//     blocks of it cost 112 registers and with them the second workgroup per CU);
//   * O^T = V^T P^T takes its V^T fragments from the row-major V tile with ds_read_b64_tr_b16 (the k order of the
//     transposed read matches the accumulator-as-operand order of P, attn_common.h::pack_half): no V^T copy, no
//     transpose kernel.
// Measured at the base_det shape (2 x 64 x 64 tokens, 12 heads, 600 workgroups; rocprofv3, tools/bench_attn_win.py):
// 25.6 us against 36.6 + 9 us (forward + transpose) of the general kernel.  Of the 25.6 us, 11.2 us remain with
// staging AND both passes removed (launch, Q loads, output stores of 600 x 448-thread workgroups over 256 CUs), 4.4 us
// is staging, 10 us the two passes: the floor, not the matrix pipe, keeps this kernel at 0.09 of the MFMA peak.
#include "attn_common.h"
#include "common.h"

namespace vah {
namespace attn {
namespace {

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;


// Stages NT (rows, 64) bf16 tensors of one window into LDS tiles of 32 * NB rows (144-byte rows): rows beyond N and
// padded tokens become zero rows.  4 pieces of 16 bytes per thread and tensor, every load issued before the first
// LDS store (one memory round trip for the whole tile).
template <int NB, int NT>
__device__ __forceinline__ void stage_rows(const __bf16 *const (&src)[NT], const int64_t (&lds)[NT], __bf16 *const (&dst)[NT],
                                           const RowMap &rm, int z, int N) {
    bf16x8 x[4][NT];
    bool ok[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 64 * NB;
        const int row = i >> 3, c = (i & 7) * 8;
        const int64_t g = row < N ? grow(rm, z, row, N) : -1;
        ok[it] = g >= 0;
        const int64_t gl = ok[it] ? g : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) x[it][t] = *reinterpret_cast<const bf16x8 *>(src[t] + gl * lds[t] + c);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 64 * NB;
        const int row = i >> 3, c = (i & 7) * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bf16x8 y = x[it][t];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = ok[it] ? y[j] : (__bf16)0.f;
            *reinterpret_cast<bf16x8 *>(dst[t] + row * kPadRow + c) = y;
        }
    }
}

// A operand T^T[m = d][k = token] of a row-major LDS tile T[token][d], in the k order of pack_half: lane 4q+p of a
// 16-lane group names token row q, d columns 4p..4p+3 of a 4 x 16 block and receives d column (lane & 15) of the 4
// tokens; group g: d columns 16(g&1).., tokens 4(g>>1) + q, the second read 8 tokens further.  `base` = tile_lane_base().
__device__ __forceinline__ const __bf16 *tile_lane_base(const __bf16 *tile, int lane) {
    const int grp = lane >> 4, i16 = lane & 15;
    return tile + (4 * (grp >> 1) + (i16 >> 2)) * kPadRow + 16 * (grp & 1) + 4 * (i16 & 3);
}
__device__ __forceinline__ bf16x8 load_tr(const __bf16 *base, int tok0, int db) {
    const __bf16 *p0 = base + tok0 * kPadRow + 32 * db;
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p0);
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p0 + 8 * kPadRow));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NB>
__global__ __launch_bounds__(64 * NB) __attribute__((amdgpu_waves_per_eu((2 * NB + 3) / 4, (2 * NB + 3) / 4))) void attn_win_fwd_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v, int64_t ld, RowMap rm,
    int N, int H, float scale_log2, __bf16 *__restrict__ out, int64_t ld_out, float *__restrict__ lse) {
    constexpr int ROWS = 32 * NB;
    __shared__ __attribute__((aligned(16))) __bf16 s_k[ROWS * kPadRow];
    __shared__ __attribute__((aligned(16))) __bf16 s_v[ROWS * kPadRow];
    const int h = blockIdx.x, z = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const __bf16 *qb = q + (int64_t)h * kHD, *kb_ = k + (int64_t)h * kHD, *vb = v + (int64_t)h * kHD;

    const int qrow = wave * 32 + r;
    const int64_t gq = qrow < N ? grow(rm, z, qrow, N) : -1;
    bf16x8 qf[4];
    {
        const int64_t gl = gq >= 0 ? gq : 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 x = *reinterpret_cast<const bf16x8 *>(qb + gl * ld + 16 * kk + 8 * hf);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[kk][j] = gq >= 0 ? x[j] : (__bf16)0.f;
        }
    }
    // ---- stage K and V: 8 x 16-byte pieces per row, zero rows for padded tokens and beyond N
    stage_rows<NB, 2>({kb_, vb}, {ld, ld}, {s_k, s_v}, rm, z, N);
    __syncthreads();

    // ---- pass 1: row maxima of S^T = K Q^T (S is not kept: holding 7 blocks of it costs 112 registers and the second
    //      workgroup per CU with them; the matrix pipe has the slack to form S twice, the VALU has none for an online
    //      rescale)
    auto s_block = [&](int kb) {
        f32x16 s = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            s = mfma(*reinterpret_cast<const bf16x8 *>(s_k + (kb * 32 + r) * kPadRow + 16 * kk + 8 * hf), qf[kk], s);
        if (kb == NB - 1) {                                   // only the last block crosses N
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kb * 32 + crow(i, hf) >= N) s[i] = -INFINITY;
        }
        return s;
    };
    float mx = -INFINITY;
#pragma unroll 1
    for (int kb = 0; kb < NB; ++kb) {
        const f32x16 s = s_block(kb);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m = mx * scale_log2;

    // ---- pass 2: P^T = exp2(S^T scale - m), O^T += V^T P^T, V^T fragments by transposed reads of the row-major V tile
    f32x16 o[2] = {zero16(), zero16()};
    float psum = 0.f;
    const __bf16 *vbase = tile_lane_base(s_v, lane);
#pragma unroll 1
    for (int kb = 0; kb < NB; ++kb) {
        f32x16 s = s_block(kb);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = __builtin_amdgcn_exp2f(s[i] * scale_log2 - m);
            psum += s[i];
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack_half(s, sp);
#pragma unroll
            for (int db = 0; db < 2; ++db) o[db] = mfma(load_tr(vbase, kb * 32 + 16 * sp, db), pf, o[db]);
        }
    }
    const float l_tot = psum + __shfl_xor(psum, 32, 64);

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
        if (hf == 0) lse[((int64_t)z * H + h) * N + qrow] = m + log2f(l_tot);
    } else if (qrow < N && hf == 0) {
        lse[((int64_t)z * H + h) * N + qrow] = INFINITY;     // padded query: p = 0 in the backward
    }
}


// ---------------------------------------------------------------------------------------
// Backward with Q, K, V, dO of the window resident (4 x 224 x 144 B = 126 KB of LDS), one workgroup per
// (window, head), NO barrier after staging.  Wave w owns query block w and key block w:
//   pass 1 (queries w):  for every key block j: S^T = K_j Q_w^T, dP^T = V_j dO_w^T, dS^T = P^T o (dP^T - delta),
//                        dQ_w^T += K_j^T dS^T                      (K_j^T by transposed LDS reads)
//   pass 2 (keys w):     for every query block i: S = Q_i K_w^T, dP = dO_i V_w^T, P, dS,
//                        dV_w^T += dO_i^T P, dK_w^T += Q_i^T dS    (dO_i^T, Q_i^T by transposed LDS reads)
// S and dP are formed twice (28 MFMAs per 32 x 32 block pair instead of 20): the alternative is to hand dS between
// waves through LDS with a barrier per block; the matrix pipe is not what bounds a 196-token window.
// delta = rowsum(dO o O) is formed here (no prologue kernel, no workspace, no transposed copies).
// ---------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(64 * NB) void attn_win_bwd_kernel(
    const __bf16 *__restrict__ q, const __bf16 *__restrict__ k, const __bf16 *__restrict__ v, int64_t ld,
    const __bf16 *__restrict__ o, const __bf16 *__restrict__ d_o, int64_t ld_out, RowMap rm, const float *__restrict__ lse,
    int N, int H, float scale, float scale_log2, __bf16 *__restrict__ dq, __bf16 *__restrict__ dk, __bf16 *__restrict__ dv,
    int64_t ld_d) {
    constexpr int ROWS = 32 * NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *s_q = reinterpret_cast<__bf16 *>(smem);
    __bf16 *s_k = s_q + ROWS * kPadRow, *s_v = s_k + ROWS * kPadRow, *s_do = s_v + ROWS * kPadRow;
    float *s_lse = reinterpret_cast<float *>(s_do + ROWS * kPadRow), *s_delta = s_lse + ROWS;
    const int h = blockIdx.x, z = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, hf = lane >> 5;
    const int row = wave * 32 + r;                      // this lane's query (pass 1) and key (pass 2)
    const int64_t grow_ = row < N ? grow(rm, z, row, N) : -1;
    const bool stored = grow_ >= 0;

    // delta of this lane's query: half a row of O from global, dO likewise (both also needed nowhere else)
    float part = 0.f;
    {
        const int64_t gl = stored ? grow_ : 0;
        const __bf16 *op = o + gl * ld_out + (int64_t)h * kHD + 32 * hf, *gp = d_o + gl * ld_out + (int64_t)h * kHD + 32 * hf;
        bf16x8 a[4], b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[c] = *reinterpret_cast<const bf16x8 *>(op + 8 * c);
            b[c] = *reinterpret_cast<const bf16x8 *>(gp + 8 * c);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)a[c][j] * (float)b[c][j];
    }
    const float lse_raw = lse[((int64_t)z * H + h) * N + (row < N ? row : 0)];
    stage_rows<NB, 4>({q + (int64_t)h * kHD, k + (int64_t)h * kHD, v + (int64_t)h * kHD, d_o + (int64_t)h * kHD},
                      {ld, ld, ld, ld_out}, {s_q, s_k, s_v, s_do}, rm, z, N);
    const float delta_q = stored ? part + __shfl_xor(part, 32, 64) : 0.f;
    const float lse_q = stored ? lse_raw : INFINITY;    // +inf: p = 0 for queries that do not exist
    if (hf == 0) {
        s_lse[row] = lse_q;
        s_delta[row] = delta_q;
    }
    __syncthreads();

    // ---------------- pass 1: dQ of query block `wave`
    {
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[kk] = *reinterpret_cast<const bf16x8 *>(s_q + row * kPadRow + 16 * kk + 8 * hf);
            dof[kk] = *reinterpret_cast<const bf16x8 *>(s_do + row * kPadRow + 16 * kk + 8 * hf);
        }
        f32x16 acc[2] = {zero16(), zero16()};
        const __bf16 *kbase = tile_lane_base(s_k, lane);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 ak = *reinterpret_cast<const bf16x8 *>(s_k + (j * 32 + r) * kPadRow + 16 * kk + 8 * hf);
                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(s_v + (j * 32 + r) * kPadRow + 16 * kk + 8 * hf);
                s = mfma(ak, qf[kk], s);
                dp = mfma(av, dof[kk], dp);
            }
#pragma unroll
            // keys beyond N need no mask: their K rows are zero in LDS, so dS^T of those rows adds nothing to dQ
            for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i] * scale_log2 - lse_q) * (dp[i] - delta_q);   // dS^T
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack_half(s, sp);
#pragma unroll
                for (int db = 0; db < 2; ++db) acc[db] = mfma(load_tr(kbase, j * 32 + 16 * sp, db), pf, acc[db]);
            }
        }
        if (stored) {
            __bf16 *op = dq + grow_ * ld_d + (int64_t)h * kHD;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 w;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) w[jj] = (__bf16)(acc[db][4 * g + jj] * scale);
                    *reinterpret_cast<bf16x4 *>(op + db * 32 + 8 * g + 4 * hf) = w;
                }
        }
    }

    // ---------------- pass 2: dK, dV of key block `wave`
    {
        // a padded key is NOT masked: it takes part in the softmax with k = v = 0 (reference quirk), only its
        // gradient has nowhere to go.  Keys beyond N need no mask either: column `key` of S only reaches dK, dV of
        // that key, which are not stored; queries that do not exist have lse = +inf (p = 0) and zero Q, dO rows.
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[kk] = *reinterpret_cast<const bf16x8 *>(s_k + row * kPadRow + 16 * kk + 8 * hf);
            vf[kk] = *reinterpret_cast<const bf16x8 *>(s_v + row * kPadRow + 16 * kk + 8 * hf);
        }
        f32x16 dkt[2] = {zero16(), zero16()}, dvt[2] = {zero16(), zero16()};
        const __bf16 *qbase = tile_lane_base(s_q, lane), *dobase = tile_lane_base(s_do, lane);
#pragma unroll 1
        for (int i = 0; i < NB; ++i) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 aq = *reinterpret_cast<const bf16x8 *>(s_q + (i * 32 + r) * kPadRow + 16 * kk + 8 * hf);
                const bf16x8 ad = *reinterpret_cast<const bf16x8 *>(s_do + (i * 32 + r) * kPadRow + 16 * kk + 8 * hf);
                s = mfma(aq, kf[kk], s);          // S[query (reg)][key (lane)]
                dp = mfma(ad, vf[kk], dp);        // dP[query][key]
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l4 = *reinterpret_cast<const float4 *>(s_lse + i * 32 + 8 * g + 4 * hf);
                const float4 d4 = *reinterpret_cast<const float4 *>(s_delta + i * 32 + 8 * g + 4 * hf);
                const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int e = 4 * g + jj;
                    const float p = __builtin_amdgcn_exp2f(s[e] * scale_log2 - ls[jj]);
                    s[e] = p;                                  // P
                    dp[e] = p * (dp[e] - de[jj]);              // dS
                }
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack_half(s, sp), dsf = pack_half(dp, sp);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dvt[db] = mfma(load_tr(dobase, i * 32 + 16 * sp, db), pf, dvt[db]);      // dV^T[d][key] += dO^T P
                    dkt[db] = mfma(load_tr(qbase, i * 32 + 16 * sp, db), dsf, dkt[db]);      // dK^T[d][key] += Q^T dS
                }
            }
        }
        if (stored) {
            __bf16 *pk = dk + grow_ * ld_d + (int64_t)h * kHD, *pv = dv + grow_ * ld_d + (int64_t)h * kHD;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 wk, wv;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        wk[jj] = (__bf16)(dkt[db][4 * g + jj] * scale);
                        wv[jj] = (__bf16)dvt[db][4 * g + jj];
                    }
                    *reinterpret_cast<bf16x4 *>(pk + db * 32 + 8 * g + 4 * hf) = wk;
                    *reinterpret_cast<bf16x4 *>(pv + db * 32 + 8 * g + 4 * hf) = wv;
                }
        }
    }
}

template <int NB>
int launch_win_bwd(const __bf16 *q, const __bf16 *k, const __bf16 *v, int64_t ld, const __bf16 *o, const __bf16 *d_o,
                   int64_t ld_out, RowMap rm, const float *lse, int64_t Z, int64_t H, int N, float scale, __bf16 *dq,
                   __bf16 *dk, __bf16 *dv, int64_t ld_d, hipStream_t st) {
    constexpr int lds = 4 * 32 * NB * kPadRow * 2 + 2 * 32 * NB * 4;
    if (int rc = allow_dynamic_lds((const void *)attn_win_bwd_kernel<NB>, lds, "attn_win_bwd")) return rc;
    hipLaunchKernelGGL((attn_win_bwd_kernel<NB>), dim3((unsigned)H, (unsigned)Z), dim3(64 * NB), lds, st, q, k, v, ld, o, d_o,
                       ld_out, rm, lse, N, (int)H, scale, scale * 1.4426950408889634f, dq, dk, dv, ld_d);
    return check_launch("attn_win_bwd");
}

template <int NB>
int launch_win_fwd(const __bf16 *q, const __bf16 *k, const __bf16 *v, int64_t ld, RowMap rm, int64_t Z, int64_t H, int N,
                   float scale_log2, __bf16 *out, int64_t ld_out, float *lse, hipStream_t st) {
    hipLaunchKernelGGL((attn_win_fwd_kernel<NB>), dim3((unsigned)H, (unsigned)Z), dim3(64 * NB), 0, st, q, k, v, ld, rm, N, (int)H,
                       scale_log2, out, ld_out, lse);
    return check_launch("attn_win_fwd");
}

}  // namespace

// Resident-window forward for N <= 224 tokens per window; returns -1 when the shape is not served.
int attn_win_fwd_resident(const void *q, const void *k, const void *v, int64_t ld, RowMap rm, int64_t Z, int64_t H, int64_t N,
                          float scale, void *out, int64_t ld_out, float *lse, hipStream_t st) {
    if (N < 1 || N > 224 || Z > 65535 || H > 65535) return -1;
    const int NB = (int)((N + 31) / 32);
    const float scale_log2 = scale * 1.4426950408889634f;
#define VAH_CASE(B_)                                                                                                    \
    if (NB == B_)                                                                                                       \
        return launch_win_fwd<B_>((const __bf16 *)q, (const __bf16 *)k, (const __bf16 *)v, ld, rm, Z, H, (int)N, scale_log2, \
                                  (__bf16 *)out, ld_out, lse, st)
    VAH_CASE(1);
    VAH_CASE(2);
    VAH_CASE(3);
    VAH_CASE(4);
    VAH_CASE(5);
    VAH_CASE(6);
    VAH_CASE(7);
#undef VAH_CASE
    return -1;
}


// Resident-window backward for N <= 224 tokens per window (no workspace, no prologue); -1 when not served.
int attn_win_bwd_resident(const void *q, const void *k, const void *v, int64_t ld, const void *o, const void *d_o,
                          int64_t ld_out, RowMap rm, const float *lse, int64_t Z, int64_t H, int64_t N, float scale, void *dq,
                          void *dk, void *dv, int64_t ld_d, hipStream_t st) {
    if (N < 1 || N > 224 || Z > 65535 || H > 65535) return -1;
    const int NB = (int)((N + 31) / 32);
#define VAH_CASE(B_)                                                                                                     \
    if (NB == B_)                                                                                                        \
        return launch_win_bwd<B_>((const __bf16 *)q, (const __bf16 *)k, (const __bf16 *)v, ld, (const __bf16 *)o,         \
                                  (const __bf16 *)d_o, ld_out, rm, lse, Z, H, (int)N, scale, (__bf16 *)dq, (__bf16 *)dk, \
                                  (__bf16 *)dv, ld_d, st)
    VAH_CASE(1);
    VAH_CASE(2);
    VAH_CASE(3);
    VAH_CASE(4);
    VAH_CASE(5);
    VAH_CASE(6);
    VAH_CASE(7);
#undef VAH_CASE
    return -1;
}

}  // namespace attn
}  // namespace vah
