// bf16 GEMMs of the Linear layers (reference: every nn.Linear of base/vit.py, adapter_modules.py and
// ops/modules/ms_deform_attn.py, i.e. F.linear and its two backward products).
//
// The tiles come from hipBLASLt (plain library GEMMs are the one place a library is the right
// tool); what lives here is the part the framework path does badly on this workload:
//   * algorithm choice.  hipBLASLt's first heuristic answer is up to 5x off for the tall-skinny
//     weight-gradient products (K = 8192..43008 rows, 768x768..3072 outputs).  Every distinct
//     problem is timed once, on first use, over the heuristic candidates (or over every algorithm
//     of the library in exhaustive mode) and the winner is cached; the table can be dumped and
//     loaded as text so a tuned table can be shipped.
//   * the weight gradient is written in fp32 straight from the accumulators (no bf16 rounding of
//     the gradient and no cast kernel per parameter).
//   * split-K.  The adapter's weight gradients reduce over 43008 token rows into outputs of at most
//     768 x 768: 18..72 output tiles for 256 CUs.  hipBLASLt runs them at 0.2 PFLOP/s; the same
//     library run as a strided batch over S slices of the rows (S x more tiles) plus one reduction
//     pass over the S fp32 partial products is 2-3x faster.  S is part of the timed choice.
// Row-major in, row-major out; hipBLASLt is column-major, so D^T = op(B)^T op(A)^T is what is run.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/vitadapter_hip.h"
#include "common.h"

namespace vah {
namespace {

struct Key {
    int ta, tb, d32, epi, bias32;
    int64_t M, N, K, lda, ldb, ldd;
    bool operator<(const Key &o) const {
        return std::tie(ta, tb, d32, epi, bias32, M, N, K, lda, ldb, ldd) <
               std::tie(o.ta, o.tb, o.d32, o.epi, o.bias32, o.M, o.N, o.K, o.lda, o.ldb, o.ldd);
    }
};

struct Choice {
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
    int index = -1;          // hipblaslt_ext algorithm index (stable within one library build)
    int split = 1;           // slices of the reduction dimension (1: hipBLASLt writes D itself)
    float us = 0.f;          // measured time of the winner (0: not measured)
    bool resolved = false;   // algo valid (an entry loaded from text is resolved on first use)
};

struct Problem;
struct State {
    std::mutex mu;
    hipblasLtHandle_t handle = nullptr;
    std::map<Key, Choice> table;
    // descriptor sets kept per problem: creating and destroying one matmul descriptor and three layouts on every
    // call was a measurable part of the ~20 us of host time a GEMM launch costs (~300 launches per training step)
    std::map<Key, std::shared_ptr<Problem>> problems;
    int mode = 1;            // 0: first heuristic answer, 1: time the heuristic candidates, 2: time all
    int candidates = 32;
    void *probe = nullptr;   // device scratch of the candidate check (two maxima)
    int rejected = 0;        // candidates dropped by the check since the library was loaded
};

State &state() {
    static State s;
    return s;
}

const char *status_name(hipblasStatus_t s) {
    switch (s) {
    case HIPBLAS_STATUS_SUCCESS: return "success";
    case HIPBLAS_STATUS_NOT_INITIALIZED: return "not initialized";
    case HIPBLAS_STATUS_ALLOC_FAILED: return "alloc failed";
    case HIPBLAS_STATUS_INVALID_VALUE: return "invalid value";
    case HIPBLAS_STATUS_NOT_SUPPORTED: return "not supported";
    case HIPBLAS_STATUS_EXECUTION_FAILED: return "execution failed";
    default: return "error";
    }
}

// RAII over the descriptor set of one problem.
struct Problem {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, ld = nullptr;
    ~Problem() {
        if (la) hipblasLtMatrixLayoutDestroy(la);
        if (lb) hipblasLtMatrixLayoutDestroy(lb);
        if (ld) hipblasLtMatrixLayoutDestroy(ld);
        if (desc) hipblasLtMatmulDescDestroy(desc);
    }
};

struct Call {
    Key k;
    const void *A, *B, *bias;
    void *D;
    void *ws;
    size_t ws_bytes;
    hipStream_t st;
    // optional finalize job riding on the reduction launch: out[c] = sum_p part[p][c] (the column-sum
    // partials of the bias gradient), so a Linear backward needs one small launch less
    const float *fin_part = nullptr;
    int fin_nparts = 0, fin_C = 0;
    float *fin_out = nullptr;
    const float *ref = nullptr;      // tuning: the reference product (M x N fp32) every candidate is held to
};

// split > 1: a strided batch over `split` equal slices of the K rows; the fp32 partial products
// (split, M, N) land in the workspace and reduce_splits() sums them into D.
hipblasStatus_t make_problem(const Call &c, Problem &p, int split) {
    const Key &k0 = c.k;
    Key k = k0;
    k.K = k0.K / split;
    if (split > 1) {
        k.d32 = 1;
        k.ldd = k0.N;
    }
    hipblasStatus_t s = hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F);
    if (s != HIPBLAS_STATUS_SUCCESS) return s;
    // column-major view: first operand = our B, second = our A
    const hipblasOperation_t op1 = k.tb ? HIPBLAS_OP_T : HIPBLAS_OP_N, op2 = k.ta ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op1, sizeof(op1));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op2, sizeof(op2));
    // our B row-major: (K x N) [tb = 0] or (N x K) [tb = 1]; as column-major (N x K) resp. (K x N)
    s = hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, k.tb ? k.K : k.N, k.tb ? k.N : k.K, k.ldb);
    if (s != HIPBLAS_STATUS_SUCCESS) return s;
    // our A row-major: (M x K) [ta = 0] or (K x M) [ta = 1]; as column-major (K x M) resp. (M x K)
    s = hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, k.ta ? k.M : k.K, k.ta ? k.K : k.M, k.lda);
    if (s != HIPBLAS_STATUS_SUCCESS) return s;
    s = hipblasLtMatrixLayoutCreate(&p.ld, k.d32 ? HIP_R_32F : HIP_R_16BF, k.N, k.M, k.ldd);
    if (s != HIPBLAS_STATUS_SUCCESS) return s;
    if (split > 1) {
        const int32_t batch = split;
        // a slice of the K rows: K/split rows further down (operand stored K-major) or K/split columns along
        const int64_t sa = k.ta ? k.K * k.lda : k.K, sb = k.tb ? k.K : k.K * k.ldb, sd = k.M * k.N;
        hipblasLtMatrixLayoutSetAttribute(p.la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch));
        hipblasLtMatrixLayoutSetAttribute(p.lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch));
        hipblasLtMatrixLayoutSetAttribute(p.ld, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch));
        hipblasLtMatrixLayoutSetAttribute(p.la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sb, sizeof(sb));
        hipblasLtMatrixLayoutSetAttribute(p.lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sa, sizeof(sa));
        hipblasLtMatrixLayoutSetAttribute(p.ld, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sd, sizeof(sd));
    }
    uint32_t epi = HIPBLASLT_EPILOGUE_DEFAULT;
    switch (k.epi) {
    case VAH_GEMM_EPI_NONE: break;
    case VAH_GEMM_EPI_BIAS: epi = HIPBLASLT_EPILOGUE_BIAS; break;
    default: return HIPBLAS_STATUS_INVALID_VALUE;
    }
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
    if (c.bias) {
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &c.bias, sizeof(c.bias));
        const int32_t bt = k.bias32 ? HIP_R_32F : HIP_R_16BF;
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
    }
    return HIPBLAS_STATUS_SUCCESS;
}

size_t partial_bytes(const Key &k, int split) {
    return split > 1 ? ((size_t)split * k.M * k.N * sizeof(float) + 255) / 256 * 256 : 0;
}

template <typename OT>
__global__ __launch_bounds__(256) void reduce_splits(const float *__restrict__ part, int split, int64_t MN, int N,
                                                     int64_t ldd, OT *__restrict__ out, int reduce_blocks,
                                                     const float *__restrict__ fin_part, int fin_nparts, int fin_C,
                                                     float *__restrict__ fin_out) {
    if ((int)blockIdx.x >= reduce_blocks) {
        // finalize job: 32 columns x 8 partial-row lanes per workgroup
        __shared__ float s_acc[8][32];
        const int col = threadIdx.x & 31, pl = threadIdx.x >> 5;
        const int k = ((int)blockIdx.x - reduce_blocks) * 32 + col;
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // 8 independent loads in flight per thread
        if (k < fin_C) {
            int p = pl;
            for (; p + 56 < fin_nparts; p += 64) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a8[u] += fin_part[(int64_t)(p + 8 * u) * fin_C + k];
            }
            for (; p < fin_nparts; p += 8) a8[0] += fin_part[(int64_t)p * fin_C + k];
        }
        s_acc[pl][col] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        __syncthreads();
        if (pl == 0 && k < fin_C) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += s_acc[u][col];
            fin_out[k] = t;
        }
        return;
    }
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < MN; i += (int64_t)reduce_blocks * 1024) {
        float4 acc = *reinterpret_cast<const float4 *>(part + i);
        for (int s = 1; s < split; ++s) {
            const float4 v = *reinterpret_cast<const float4 *>(part + s * MN + i);
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
            acc.w += v.w;
        }
        const int64_t row = i / N;
        OT *o = out + row * ldd + (i - row * N);
        o[0] = (OT)acc.x;
        o[1] = (OT)acc.y;
        o[2] = (OT)acc.z;
        o[3] = (OT)acc.w;
    }
}

// ---- numerical check of a tuning candidate: its WHOLE output against a plain fp32-accumulating product of the same
// operands computed here (round 2 found exhaustive-mode algorithms that run without an error status and return wrong
// numbers; round 3 found heuristic answers for 1024 x N x 64 that leave most of the output unwritten once the workspace
// holds another product's data - and whose first 64 columns are right).  So the candidate runs twice on an output
// filled with NaN patterns, first over a workspace filled with 0x01 bytes, then over whatever it left there, and every element
// of both results has to agree with the reference.
__global__ __launch_bounds__(256) void ref_gemm(int ta, int tb, int M, int N, int K, const __bf16 *__restrict__ A, int64_t lda,
                                                const __bf16 *__restrict__ B, int64_t ldb, const void *__restrict__ bias,
                                                int bias32, float *__restrict__ ref) {
    __shared__ float sa[16][65], sb[16][65];
    const int tm = blockIdx.y * 64, tn = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 1024; i += 256) {
            // the operand's contiguous dimension runs along the lanes
            const int ma = ta ? (i & 63) : (i >> 4), ka = ta ? (i >> 6) : (i & 15);
            const int m = tm + ma, k = k0 + ka;
            sa[ka][ma] = (m < M && k < K) ? (float)(ta ? A[(int64_t)k * lda + m] : A[(int64_t)m * lda + k]) : 0.f;
            const int nb = tb ? (i >> 4) : (i & 63), kb = tb ? (i & 15) : (i >> 6);
            const int n = tn + nb, k2 = k0 + kb;
            sb[kb][nb] = (n < N && k2 < K) ? (float)(tb ? B[(int64_t)n * ldb + k2] : B[(int64_t)k2 * ldb + n]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = sa[kk][ty * 4 + i];
                b[i] = sb[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const int m = tm + ty * 4 + i, n = tn + tx * 4 + j;
            if (m >= M || n >= N) continue;
            float v = acc[i][j];
            if (bias) v += bias32 ? ((const float *)bias)[n] : (float)((const __bf16 *)bias)[n];
            ref[(int64_t)m * N + n] = v;
        }
}

template <typename OT>
__global__ __launch_bounds__(256) void full_compare(const OT *__restrict__ D, int64_t ldd, int64_t MN, int N,
                                                    const float *__restrict__ ref, unsigned *__restrict__ out2) {
    float md = 0.f, mr = 0.f;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < MN; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N;
        const float v = (float)D[r * ldd + (i - r * N)], w = ref[i];
        const float d = fabsf(v - w);
        bad = bad || !(d == d);
        md = fmaxf(md, d == d ? d : 0.f);
        mr = fmaxf(mr, fabsf(w));
    }
    // non-negative floats order like their bit patterns
    if (bad || md > 0.f) atomicMax(out2, __float_as_uint(bad ? INFINITY : md));
    if (mr > 0.f) atomicMax(out2 + 1, __float_as_uint(mr));
}

// The reference product of this call as M x N fp32 (device memory the caller frees), nullptr if it cannot be had.
float *make_reference(const Call &c) {
    const Key &k = c.k;
    if (k.M > (64 << 16) || k.N > (int64_t)INT32_MAX || k.K > (int64_t)INT32_MAX) return nullptr;
    float *ref = nullptr;
    if (hipMalloc((void **)&ref, (size_t)k.M * k.N * sizeof(float)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(ref_gemm, dim3((unsigned)((k.N + 63) / 64), (unsigned)((k.M + 63) / 64)), dim3(256), 0, c.st, k.ta, k.tb,
                       (int)k.M, (int)k.N, (int)k.K, (const __bf16 *)c.A, k.lda, (const __bf16 *)c.B, k.ldb, c.bias, k.bias32, ref);
    if (hipGetLastError() != hipSuccess) {
        (void)hipFree(ref);
        return nullptr;
    }
    return ref;
}

hipblasStatus_t run(State &S, const Call &c, Problem &p, const hipblasLtMatmulAlgo_t &algo, size_t ws_need, int split);

// -> true if the algorithm reproduces the reference on NaN-filled output, over a poisoned and over a used workspace
bool validate(State &S, const Call &c, Problem &p, const hipblasLtMatmulAlgo_t &algo, size_t ws_need, int split, const float *ref) {
    if (!ref) return true;                                                                           // cannot check: accept
    if (!S.probe && hipMalloc(&S.probe, 2 * sizeof(unsigned)) != hipSuccess) return true;
    unsigned *out2 = (unsigned *)S.probe;
    const Key &k = c.k;
    const size_t es = k.d32 ? 4 : 2;
    const int64_t MN = k.M * k.N;
    const unsigned blocks = (unsigned)std::min<int64_t>(4096, (MN + 255) / 256);
    // dirty, but not hostile: 0x01 bytes are small positive integers / denormal floats - a kernel that expects zeroed flags
    // or partial sums goes wrong on them (and is refused below) without being handed 0xFFFFFFFF as an index
    if (c.ws_bytes && hipMemsetAsync(c.ws, 0x01, c.ws_bytes, c.st) != hipSuccess) return true;
    for (int pass = 0; pass < 2; ++pass) {
        if (hipMemsetAsync(out2, 0, 8, c.st) != hipSuccess) return true;
        if (hipMemset2DAsync(c.D, (size_t)k.ldd * es, 0xFF, (size_t)k.N * es, (size_t)k.M, c.st) != hipSuccess) return true;
        if (run(S, c, p, algo, ws_need, split) != HIPBLAS_STATUS_SUCCESS) return false;
        if (k.d32)
            hipLaunchKernelGGL(full_compare<float>, dim3(blocks), dim3(256), 0, c.st, (const float *)c.D, k.ldd, MN, (int)k.N, ref, out2);
        else
            hipLaunchKernelGGL(full_compare<__bf16>, dim3(blocks), dim3(256), 0, c.st, (const __bf16 *)c.D, k.ldd, MN, (int)k.N, ref, out2);
        float h[2] = {0.f, 0.f};
        if (hipMemcpyAsync(h, out2, 8, hipMemcpyDeviceToHost, c.st) != hipSuccess || hipStreamSynchronize(c.st) != hipSuccess) return true;
        // same operands, fp32 accumulation in a different order, bf16 or fp32 result: 2^-6 of the largest value is far
        // above any legitimate difference and far below a wrong or missing tile
        if (!(h[0] <= 0.015625f * fmaxf(h[1], 1e-20f))) return false;
    }
    return true;
}

hipblasStatus_t run(State &S, const Call &c, Problem &p, const hipblasLtMatmulAlgo_t &algo, size_t ws_need, int split) {
    const size_t pb = partial_bytes(c.k, split);
    if (pb + ws_need > c.ws_bytes) return HIPBLAS_STATUS_ALLOC_FAILED;
    const float alpha = 1.f, beta = 0.f;
    void *d = split > 1 ? c.ws : c.D;
    // operands swapped: see the header comment
    const hipblasStatus_t s = hipblasLtMatmul(S.handle, p.desc, &alpha, c.B, p.la, c.A, p.lb, &beta, d, p.ld, d, p.ld, &algo,
                                              (char *)c.ws + pb, c.ws_bytes - pb, c.st);
    if (s != HIPBLAS_STATUS_SUCCESS) return s;
    const bool fin = c.fin_part != nullptr;
    if (split == 1 && !fin) return s;
    const int64_t MN = c.k.M * c.k.N;
    const unsigned rblocks = split > 1 ? (unsigned)std::min<int64_t>(2048, (MN / 4 + 255) / 256) : 0u;
    const unsigned fblocks = fin ? (unsigned)((c.fin_C + 31) / 32) : 0u;
    if (c.k.d32)
        hipLaunchKernelGGL(reduce_splits<float>, dim3(rblocks + fblocks), dim3(256), 0, c.st, (const float *)c.ws, split, MN,
                           (int)c.k.N, c.k.ldd, (float *)c.D, (int)rblocks, c.fin_part, c.fin_nparts, c.fin_C, c.fin_out);
    else
        hipLaunchKernelGGL(reduce_splits<__bf16>, dim3(rblocks + fblocks), dim3(256), 0, c.st, (const float *)c.ws, split, MN,
                           (int)c.k.N, c.k.ldd, (__bf16 *)c.D, (int)rblocks, c.fin_part, c.fin_nparts, c.fin_C, c.fin_out);
    return HIPBLAS_STATUS_SUCCESS;
}

// Time one candidate on the caller's stream (the output is simply overwritten).
float time_algo(State &S, const Call &c, Problem &p, const hipblasLtMatmulAlgo_t &algo, size_t ws_need, int split,
                hipEvent_t e0, hipEvent_t e1, int reps) {
    if (run(S, c, p, algo, ws_need, split) != HIPBLAS_STATUS_SUCCESS) return -1.f;      // warm-up + validity
    if (hipEventRecord(e0, c.st) != hipSuccess) return -1.f;
    for (int i = 0; i < reps; ++i)
        if (run(S, c, p, algo, ws_need, split) != HIPBLAS_STATUS_SUCCESS) return -1.f;
    if (hipEventRecord(e1, c.st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
    return ms * 1000.f / reps;
}

// Best algorithm for one value of `split` (timed unless there is a single candidate and nothing to
// compare it with).
hipblasStatus_t choose_for_split(State &S, const Call &c, Problem &p, int split, bool must_time, Choice &out) {
    const size_t pb = partial_bytes(c.k, split);
    if (pb >= c.ws_bytes) return HIPBLAS_STATUS_ALLOC_FAILED;
    const size_t ws_lib = c.ws_bytes - pb;
    std::vector<hipblasLtMatmulHeuristicResult_t> cand;
    const bool d32 = split > 1 || c.k.d32;
    if (S.mode == 2) {
        std::vector<hipblasLtMatmulHeuristicResult_t> all;
        const Key &k = c.k;
        hipblasStatus_t s = hipblaslt_ext::getAllAlgos(
            S.handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, k.tb ? HIPBLAS_OP_T : HIPBLAS_OP_N,
            k.ta ? HIPBLAS_OP_T : HIPBLAS_OP_N, HIP_R_16BF, HIP_R_16BF, d32 ? HIP_R_32F : HIP_R_16BF,
            d32 ? HIP_R_32F : HIP_R_16BF, HIPBLAS_COMPUTE_32F, all);
        if (s == HIPBLAS_STATUS_SUCCESS)
            for (auto &r : all) {
                size_t need = 0;
                const float one = 1.f, zero = 0.f;
                if (hipblaslt_ext::matmulIsAlgoSupported(S.handle, p.desc, &one, p.la, p.lb, &zero, p.ld, p.ld, r.algo,
                                                         need) == HIPBLAS_STATUS_SUCCESS &&
                    need <= ws_lib) {
                    r.workspaceSize = need;
                    cand.push_back(r);
                }
            }
    }
    if (cand.empty()) {
        hipblasLtMatmulPreference_t pref = nullptr;
        hipblasStatus_t s = hipblasLtMatmulPreferenceCreate(&pref);
        if (s != HIPBLAS_STATUS_SUCCESS) return s;
        const uint64_t wsmax = ws_lib;
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax));
        const int want = S.mode == 0 ? 1 : (split > 1 ? std::min(S.candidates, 12) : S.candidates);
        cand.resize(want);
        int got = 0;
        s = hipblasLtMatmulAlgoGetHeuristic(S.handle, p.desc, p.la, p.lb, p.ld, p.ld, pref, want, cand.data(), &got);
        hipblasLtMatmulPreferenceDestroy(pref);
        if (s != HIPBLAS_STATUS_SUCCESS) return s;
        cand.resize(std::max(got, 0));
        if (cand.empty()) return HIPBLAS_STATUS_NOT_SUPPORTED;
    }
    out.split = split;
    if (S.mode == 0 || (cand.size() == 1 && !must_time)) {
        if (!validate(S, c, p, cand[0].algo, cand[0].workspaceSize, split, c.ref)) {
            ++S.rejected;
            return HIPBLAS_STATUS_EXECUTION_FAILED;
        }
        out.algo = cand[0].algo;
        out.workspace = cand[0].workspaceSize;
        out.index = hipblaslt_ext::getIndexFromAlgo(out.algo);
        out.resolved = true;
        return HIPBLAS_STATUS_SUCCESS;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return HIPBLAS_STATUS_ALLOC_FAILED;
    float best = -1.f;
    size_t best_i = 0;
    // coarse pass over everything, then a longer look at the fastest few THAT COMPUTE THE REFERENCE's NUMBERS (walked in
    // order of speed: an algorithm that skips part of the output is fast)
    std::vector<std::pair<float, size_t>> timed;
    for (size_t i = 0; i < cand.size(); ++i) {
        const float us = time_algo(S, c, p, cand[i].algo, cand[i].workspaceSize, split, e0, e1, 3);
        if (us > 0.f) timed.emplace_back(us, i);
    }
    std::sort(timed.begin(), timed.end());
    int looked = 0;
    for (size_t j = 0; j < timed.size() && looked < 4; ++j) {
        const size_t i = timed[j].second;
        if (!validate(S, c, p, cand[i].algo, cand[i].workspaceSize, split, c.ref)) {
            ++S.rejected;
            continue;
        }
        ++looked;
        const float us = time_algo(S, c, p, cand[i].algo, cand[i].workspaceSize, split, e0, e1, 20);
        if (us > 0.f && (best < 0.f || us < best)) {
            best = us;
            best_i = i;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (best < 0.f) return HIPBLAS_STATUS_EXECUTION_FAILED;
    out.algo = cand[best_i].algo;
    out.workspace = cand[best_i].workspaceSize;
    out.index = hipblaslt_ext::getIndexFromAlgo(out.algo);
    out.us = best;
    out.resolved = true;
    return HIPBLAS_STATUS_SUCCESS;
}

// Slices worth trying: only reductions that are long and leave most of the chip without an output
// tile; the partial products must fit the workspace next to the library's own scratch.
std::vector<int> split_candidates(const State &S, const Call &c) {
    std::vector<int> v{1};
    const Key &k = c.k;
    if (S.mode == 0 || c.bias || k.K < 4096 || k.N % 4) return v;
    const int64_t tiles = ((k.M + 127) / 128) * ((k.N + 127) / 128);
    if (tiles >= 256) return v;
    for (int s = 2; s <= 64; s *= 2) {
        if (k.K % s || k.K / s < 512 || tiles * s > 4096) break;
        if (partial_bytes(k, s) + (8u << 20) > c.ws_bytes) break;
        v.push_back(s);
    }
    return v;
}

hipblasStatus_t choose(State &S, const Call &call, Choice &out) {
    Call c = call;                       // candidates are timed without the piggy-backed finalize job
    c.fin_part = nullptr;
    const std::vector<int> splits = split_candidates(S, c);
    float *ref = S.mode != 0 ? make_reference(c) : nullptr;
    c.ref = ref;
    hipblasStatus_t last = HIPBLAS_STATUS_NOT_SUPPORTED;
    bool have = false;
    for (int split : splits) {
        Problem p;
        hipblasStatus_t s = make_problem(c, p, split);
        Choice ch;
        if (s == HIPBLAS_STATUS_SUCCESS) s = choose_for_split(S, c, p, split, splits.size() > 1, ch);
        if (s != HIPBLAS_STATUS_SUCCESS) {
            last = s;
            continue;
        }
        if (!have || (ch.us > 0.f && ch.us < out.us)) out = ch;
        have = true;
    }
    if (ref) {
        (void)hipStreamSynchronize(c.st);
        (void)hipFree(ref);
    }
    return have ? HIPBLAS_STATUS_SUCCESS : last;
}

// An entry that came from text: look the algorithm up by index and make sure it fits this problem.
bool resolve(State &S, const Call &c, Problem &p, Choice &ch) {
    std::vector<int> idx{ch.index};
    std::vector<hipblasLtMatmulHeuristicResult_t> res;
    if (ch.index < 0 || hipblaslt_ext::getAlgosFromIndex(S.handle, idx, res) != HIPBLAS_STATUS_SUCCESS || res.empty())
        return false;
    size_t need = 0;
    const float one = 1.f, zero = 0.f;
    if (hipblaslt_ext::matmulIsAlgoSupported(S.handle, p.desc, &one, p.la, p.lb, &zero, p.ld, p.ld, res[0].algo, need) !=
            HIPBLAS_STATUS_SUCCESS || need + partial_bytes(c.k, ch.split) > c.ws_bytes)
        return false;
    // an entry is only as good as the build and the workspace use it was tuned with: hold it to the reference once
    Call cc = c;
    cc.fin_part = nullptr;
    float *ref = make_reference(cc);
    const bool ok = validate(S, cc, p, res[0].algo, need, ch.split, ref);
    if (ref) {
        (void)hipStreamSynchronize(c.st);
        (void)hipFree(ref);
    }
    if (!ok) {
        ++S.rejected;
        return false;
    }
    ch.algo = res[0].algo;
    ch.workspace = need;
    ch.resolved = true;
    return true;
}

}  // namespace
}  // namespace vah

extern "C" {

int vah_gemm_set_tuning(int mode, int candidates) {
    using namespace vah;
    clear_error();
    if (mode < 0 || mode > 2 || candidates < 1 || candidates > 4096) return fail(VAH_E_SHAPE, "vah_gemm_set_tuning: bad arguments");
    State &S = state();
    std::lock_guard<std::mutex> lock(S.mu);
    S.mode = mode;
    S.candidates = candidates;
    return VAH_OK;
}

static int gemm_impl(const char *fn, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                     const void *B, int64_t ldb, void *D, int64_t ldd, int d_is_f32, int epilogue, const void *bias,
                     int bias_is_f32, void *workspace, int64_t workspace_bytes, void *stream, const float *fin_part,
                     int64_t fin_nparts, int64_t fin_C, float *fin_out);

int vah_gemm_bf16(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                  const void *B, int64_t ldb, void *D, int64_t ldd, int d_is_f32, int epilogue, const void *bias,
                  int bias_is_f32, void *workspace, int64_t workspace_bytes, void *stream) {
    return gemm_impl("vah_gemm_bf16", trans_a, trans_b, M, N, K, A, lda, B, ldb, D, ldd, d_is_f32, epilogue, bias, bias_is_f32,
                     workspace, workspace_bytes, stream, nullptr, 0, 0, nullptr);
}

// vah_gemm_bf16 plus a finalize job on its last launch: fin_out[c] = sum_{p < fin_nparts} fin_part[p * fin_C + c]
// (the partial rows vah_colsum_bf16_partials wrote).  One Linear backward = partials, GEMM, reduction +
// finalize: a launch less than column sum and weight gradient on their own.
int vah_gemm_bf16_fin(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                      const void *B, int64_t ldb, void *D, int64_t ldd, int d_is_f32, void *workspace,
                      int64_t workspace_bytes, const float *fin_part, int64_t fin_nparts, int64_t fin_C, float *fin_out,
                      void *stream) {
    using namespace vah;
    if (!fin_part || !fin_out || fin_nparts < 1 || fin_C < 1 || fin_nparts > (1 << 20) || fin_C > (1 << 24)) {
        clear_error();
        return fail(VAH_E_SHAPE, "vah_gemm_bf16_fin: bad finalize job");
    }
    return gemm_impl("vah_gemm_bf16_fin", trans_a, trans_b, M, N, K, A, lda, B, ldb, D, ldd, d_is_f32, VAH_GEMM_EPI_NONE, nullptr,
                     0, workspace, workspace_bytes, stream, fin_part, fin_nparts, fin_C, fin_out);
}

static int gemm_impl(const char *fn, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                     const void *B, int64_t ldb, void *D, int64_t ldd, int d_is_f32, int epilogue, const void *bias,
                     int bias_is_f32, void *workspace, int64_t workspace_bytes, void *stream, const float *fin_part,
                     int64_t fin_nparts, int64_t fin_C, float *fin_out) {
    using namespace vah;
    clear_error();
    if (M < 0 || N < 0 || K < 0) return fail(VAH_E_SHAPE, "%s: negative dimension", fn);
    if (M == 0 || N == 0) return VAH_OK;
    if (K == 0) return fail(VAH_E_SHAPE, "%s: K = 0 (zero-fill the output instead)", fn);
    if (!A || !B || !D) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldd < N) return fail(VAH_E_SHAPE, "%s: leading dimension too small", fn);
    if (epilogue != VAH_GEMM_EPI_NONE && epilogue != VAH_GEMM_EPI_BIAS) return fail(VAH_E_SHAPE, "%s: unknown epilogue", fn);
    if ((epilogue == VAH_GEMM_EPI_BIAS) != (bias != nullptr)) return fail(VAH_E_NULL, "%s: bias does not match the epilogue", fn);
    if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace)) return fail(VAH_E_NULL, "%s: workspace", fn);

    State &S = state();
    std::lock_guard<std::mutex> lock(S.mu);
    if (!S.handle) {
        const hipblasStatus_t s = hipblasLtCreate(&S.handle);
        if (s != HIPBLAS_STATUS_SUCCESS) return fail(VAH_E_UNSUPPORTED, "%s: hipblasLtCreate: %s", fn, status_name(s));
    }
    Call c{{trans_a ? 1 : 0, trans_b ? 1 : 0, d_is_f32 ? 1 : 0, epilogue, bias_is_f32 ? 1 : 0, M, N, K, lda, ldb, ldd},
           A, B, bias, D, workspace, (size_t)workspace_bytes, (hipStream_t)stream};
    c.fin_part = fin_part;
    c.fin_nparts = (int)fin_nparts;
    c.fin_C = (int)fin_C;
    c.fin_out = fin_out;
    hipblasStatus_t s = HIPBLAS_STATUS_SUCCESS;
    // named by the product's role in a Linear layer: nt = forward (x W^T), nn = input gradient (g W), tn = weight gradient
    // (g^T x); flops for the MFMA roofline rows of bench.py
    const char *role = trans_a ? (fin_part ? "gemm_tn_fin" : "gemm_tn") : (trans_b ? "gemm_nt" : "gemm_nn");
    LaunchScope scope(role, (M * K + K * N) * 2 + M * N * (d_is_f32 ? 4 : 2), c.st, 0, 2 * M * N * K);
    auto it = S.table.find(c.k);
    if (it != S.table.end() && !it->second.resolved) {
        Problem p;
        const int split = it->second.split;
        if (split < 1 || K % split || make_problem(c, p, split) != HIPBLAS_STATUS_SUCCESS ||
            partial_bytes(c.k, split) >= c.ws_bytes || !resolve(S, c, p, it->second)) {
            S.table.erase(it);
            S.problems.erase(c.k);
            it = S.table.end();
        }
    }
    if (it == S.table.end()) {
        Choice ch;
        s = choose(S, c, ch);
        if (s != HIPBLAS_STATUS_SUCCESS)
            return fail(VAH_E_UNSUPPORTED, "%s: no algorithm for %lldx%lldx%lld ta=%d tb=%d f32=%d epi=%d: %s", fn,
                        (long long)M, (long long)N, (long long)K, trans_a, trans_b, d_is_f32, epilogue, status_name(s));
        it = S.table.emplace(c.k, ch).first;
    }
    std::shared_ptr<Problem> &pp = S.problems[c.k];
    if (!pp) {
        pp = std::make_shared<Problem>();
        s = make_problem(c, *pp, it->second.split);
        if (s != HIPBLAS_STATUS_SUCCESS) {
            S.problems.erase(c.k);
            return fail(VAH_E_UNSUPPORTED, "%s: descriptor: %s", fn, status_name(s));
        }
    } else if (c.bias) {                    // the cached descriptor carries the previous call's bias pointer
        hipblasLtMatmulDescSetAttribute(pp->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &c.bias, sizeof(c.bias));
    }
    Problem &p = *pp;
    s = run(S, c, p, it->second.algo, it->second.workspace, it->second.split);
    if (s != HIPBLAS_STATUS_SUCCESS) return fail(VAH_E_UNSUPPORTED, "%s: hipblasLtMatmul: %s", fn, status_name(s));
    return check_launch(fn);
}

// Version of the hipBLASLt build behind the dispatcher (algorithm indices of a dumped table are only
// meaningful for the same build); < 0 on error.
int64_t vah_gemm_rejected_candidates(void) {
    vah::State &S = vah::state();
    std::lock_guard<std::mutex> lock(S.mu);
    return S.rejected;
}

int64_t vah_gemm_library_version(void) {
    using namespace vah;
    clear_error();
    State &S = state();
    std::lock_guard<std::mutex> lock(S.mu);
    if (!S.handle && hipblasLtCreate(&S.handle) != HIPBLAS_STATUS_SUCCESS)
        return fail(VAH_E_UNSUPPORTED, "vah_gemm_library_version: hipblasLtCreate failed");
    int v = 0;
    if (hipblasLtGetVersion(S.handle, &v) != HIPBLAS_STATUS_SUCCESS)
        return fail(VAH_E_UNSUPPORTED, "vah_gemm_library_version: hipblasLtGetVersion failed");
    return v;
}

// One line per problem: "ta tb d32 epi bias32 M N K lda ldb ldd index split us".
int64_t vah_gemm_table_dump(char *buf, int64_t cap) {
    using namespace vah;
    State &S = state();
    std::lock_guard<std::mutex> lock(S.mu);
    std::ostringstream os;
    for (auto &kv : S.table) {
        const Key &k = kv.first;
        os << k.ta << ' ' << k.tb << ' ' << k.d32 << ' ' << k.epi << ' ' << k.bias32 << ' ' << k.M << ' ' << k.N << ' '
           << k.K << ' ' << k.lda << ' ' << k.ldb << ' ' << k.ldd << ' ' << kv.second.index << ' ' << kv.second.split << ' '
           << kv.second.us << '\n';
    }
    const std::string t = os.str();
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(t.size(), (size_t)cap - 1);
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return (int64_t)t.size() + 1;
}

int vah_gemm_table_load(const char *text) {
    using namespace vah;
    clear_error();
    if (!text) return fail(VAH_E_NULL, "vah_gemm_table_load: null text");
    State &S = state();
    std::lock_guard<std::mutex> lock(S.mu);
    S.problems.clear();             // their split may change with the loaded choices
    std::istringstream is(text);
    std::string line;
    int n = 0;
    while (std::getline(is, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        Key k;
        Choice ch;
        if (!(ls >> k.ta >> k.tb >> k.d32 >> k.epi >> k.bias32 >> k.M >> k.N >> k.K >> k.lda >> k.ldb >> k.ldd >> ch.index >> ch.split >> ch.us))
            return fail(VAH_E_SHAPE, "vah_gemm_table_load: malformed line '%s'", line.c_str());
        ch.resolved = false;
        S.table[k] = ch;
        ++n;
    }
    return n;
}

}  // extern "C"
