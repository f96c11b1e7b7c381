// BEiT's relative position bias around the attention kernels of attn_flash.hip.
//
// Reference (segmentation/mmseg_custom/models/backbones/base/beit.py:120-131): every block gathers
// bias[h][i][j] = table[index[i][j]][h] (N x N x heads, N = 1 + Wh * Ww with the class token) and adds it to the scores;
// autograd scatters the (heads, N, N) gradient back into the (T, heads) table.  Done with torch ops around the MFMA
// kernels that was 126 ms of a 187 ms BEiT-L step (gather 26, transposed bf16 copies 55, indexing_backward 45).
//   relpos_build: table, index -> the two operands the attention kernels read: bias * log2(e) as bf16 (heads, N, ldb)
//                 and its per-head transpose, both written with coalesced rows (two passes over the index);
//   relpos_grad:  dS (B, heads, N, ldb) bf16 (written per image by the dQ kernel) -> d table (T, heads): one workgroup
//                 per (head, slab of query rows) adds its elements into LDS bins (consecutive keys of a row hit
//                 consecutive bins, so the lanes of a wave do not collide), the slabs' bins are summed by a second
//                 launch.  The sum order inside a slab is the LDS unit's: results are reproducible to fp32 rounding,
//                 not bitwise.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace vah {
namespace {

constexpr float kLog2e = 1.4426950408889634f;

// TRANSPOSED == false: out[h][i][j] = table[index[i][j]][h] * log2e;  true: out[h][j][i] (rows = keys)
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void relpos_build_kernel(const float *__restrict__ table, const int64_t *__restrict__ index, int N,
                                                           int H, int64_t ldb, __bf16 *__restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;        // out[h][row][col]
    if (col >= ldb) return;
    if (col >= N) {                      // padding columns: zero (the kernels read them in their last tile)
        for (int h = 0; h < H; ++h) out[((int64_t)h * N + row) * ldb + col] = (__bf16)0.f;
        return;
    }
    const int64_t t = TRANSPOSED ? index[(int64_t)col * N + row] : index[(int64_t)row * N + col];
    const float *tp = table + t * H;
    for (int h = 0; h < H; ++h) out[((int64_t)h * N + row) * ldb + col] = (__bf16)(tp[h] * kLog2e);
}

__global__ __launch_bounds__(256) void relpos_grad_kernel(const __bf16 *__restrict__ ds, const int64_t *__restrict__ index, int B,
                                                          int H, int N, int64_t ldb, int T, int rows_per_slab,
                                                          float *__restrict__ part) {
    extern __shared__ float s_bins[];
    const int h = blockIdx.y, slab = blockIdx.x;
    for (int t = threadIdx.x; t < T; t += 256) s_bins[t] = 0.f;
    __syncthreads();
    const int i0 = slab * rows_per_slab, i1 = min(N, i0 + rows_per_slab);
    for (int i = i0; i < i1; ++i)
        for (int j = threadIdx.x; j < N; j += 256) {
            const int t = (int)index[(int64_t)i * N + j];
            float v = 0.f;
            for (int b = 0; b < B; ++b) v += (float)ds[(((int64_t)b * H + h) * N + i) * ldb + j];
            atomicAdd(&s_bins[t], v);
        }
    __syncthreads();
    float *pp = part + ((int64_t)slab * H + h) * T;
    for (int t = threadIdx.x; t < T; t += 256) pp[t] = s_bins[t];
}

// dtable[t][h] = sum over slabs of part[slab][h][t]
__global__ __launch_bounds__(256) void relpos_grad_reduce(const float *__restrict__ part, int slabs, int H, int T,
                                                          float *__restrict__ dtable) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)T * H) return;
    const int h = (int)(i % H), t = (int)(i / H);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int sl = 0;
    for (; sl + 3 < slabs; sl += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += part[((int64_t)(sl + u) * H + h) * T + t];
    }
    for (; sl < slabs; ++sl) s[0] += part[((int64_t)sl * H + h) * T + t];
    dtable[i] = (s[0] + s[1]) + (s[2] + s[3]);
}

constexpr int kSlabs = 32;

}  // namespace
}  // namespace vah

extern "C" {

int vah_relpos_bias_build(const float *table, const int64_t *index, int64_t T, int64_t H, int64_t N, int64_t ldb, void *bias,
                          void *bias_t, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_relpos_bias_build";
    if (T < 1 || H < 1 || N < 1 || ldb < N || N > 65535) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (!table || !index || !bias || !bias_t) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)stream;
    LaunchScope scope("relpos_bias_build", 2 * N * N * 8 + 2 * H * N * N * 2, st);
    const dim3 grid((unsigned)((ldb + 255) / 256), (unsigned)N);
    hipLaunchKernelGGL(relpos_build_kernel<false>, grid, dim3(256), 0, st, table, index, (int)N, (int)H, ldb, (__bf16 *)bias);
    hipLaunchKernelGGL(relpos_build_kernel<true>, grid, dim3(256), 0, st, table, index, (int)N, (int)H, ldb, (__bf16 *)bias_t);
    return check_launch(fn);
}

int64_t vah_relpos_bias_grad_ws_floats(int64_t T, int64_t H) { return vah::kSlabs * H * T; }

int vah_relpos_bias_grad(const void *ds, const int64_t *index, int64_t B, int64_t H, int64_t N, int64_t ldb, int64_t T, float *ws,
                         float *dtable, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_relpos_bias_grad";
    if (T < 1 || H < 1 || N < 1 || B < 1 || ldb < N || H > 65535 || T * 4 > 150 * 1024) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (!ds || !index || !ws || !dtable) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)stream;
    const int rows_per_slab = (int)((N + kSlabs - 1) / kSlabs), slabs = (int)((N + rows_per_slab - 1) / rows_per_slab);
    const int lds = (int)(T * 4);
    if (int rc = allow_dynamic_lds((const void *)relpos_grad_kernel, lds, fn)) return rc;
    LaunchScope scope("relpos_bias_grad", B * H * N * N * 2 + N * N * 8, st);
    hipLaunchKernelGGL(relpos_grad_kernel, dim3(slabs, (unsigned)H), dim3(256), lds, st, (const __bf16 *)ds, index, (int)B, (int)H, (int)N,
                       ldb, (int)T, rows_per_slab, ws);
    if (int rc = check_launch(fn)) return rc;
    hipLaunchKernelGGL(relpos_grad_reduce, dim3((unsigned)((T * H + 255) / 256)), dim3(256), 0, st, (const float *)ws, slabs, (int)H, (int)T,
                       dtable);
    return check_launch(fn);
}

}  // extern "C"
