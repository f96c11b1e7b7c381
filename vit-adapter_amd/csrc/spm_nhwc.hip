// Memory-bound operators of the SpatialPriorModule on NHWC bf16 activations - the layout the implicit-GEMM
// convolutions of conv.hip read and write.  Reference: the conv -> SyncBatchNorm -> ReLU triples and the max-pool
// of /root/reference/detection/mmdet_custom/models/backbones/adapter_modules.py:217-260.
//   * image (N, 3, H, W) fp32 -> (N, H, W, 16) bf16 (channels 3..15 zero: the first conv reads 16-channel pixels);
//   * BatchNorm statistics: per-channel sum and sum of squares of a (rows, C) matrix - per-workgroup partial rows in a
//     workspace, summed in a fixed order by the second launch (no atomics);
//   * normalise + ReLU; backward statistics (sum g', sum g' xhat with g' = dy where the output was positive) and
//     the input gradient;
//   * MaxPool2d(3, stride 2, padding 1) with the window position of the first maximum kept per output element, and its
//     gathering backward.
// Every kernel moves 16-byte pieces (8 channels of one pixel); a thread keeps the same 8 channels for all its pieces,
// so per-channel parameters are loaded once.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "attn_common.h"
#include "common.h"

namespace vah {
namespace {

using attn::bf16x8;

constexpr int kStatParts = 512;         // partial rows of the statistics passes (two workgroups per CU)

__global__ __launch_bounds__(256) void image_to_nhwc16_kernel(const float *__restrict__ x, int64_t HW, int64_t total,
                                                              __bf16 *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // pixel over (n, hw)
    if (i >= total) return;
    const int64_t n = i / HW, p = i - n * HW;
    const float *px = x + n * 3 * HW + p;
    bf16x8 a, z;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = z[j] = (__bf16)0.f;
    a[0] = (__bf16)px[0];
    a[1] = (__bf16)px[HW];
    a[2] = (__bf16)px[2 * HW];
    *reinterpret_cast<bf16x8 *>(y + i * 16) = a;
    *reinterpret_cast<bf16x8 *>(y + i * 16 + 8) = z;
}

// image (N, C, H, W) fp32 -> patch rows (N * H/ps * W/ps, C * ps * ps) bf16, column = (c, ky, kx): the operand of the
// patch embedding as a GEMM (a ps x ps / stride ps convolution is a matrix product of the patch rows with the
// flattened filters).  One thread: 8 consecutive kx of one (patch, c, ky).
__global__ __launch_bounds__(256) void patchify_kernel(const float *__restrict__ x, int C, int H, int W, int ps, int64_t items,
                                                       __bf16 *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const int k8 = ps >> 3, Wp = W / ps, Hp = H / ps;
    int64_t t = i;
    const int kx8 = (int)(t % k8);
    t /= k8;
    const int ky = (int)(t % ps);
    t /= ps;
    const int c = (int)(t % C);
    t /= C;
    const int px = (int)(t % Wp);
    t /= Wp;
    const int py = (int)(t % Hp), n = (int)(t / Hp);
    const float *src = x + (((int64_t)n * C + c) * H + py * ps + ky) * W + px * ps + kx8 * 8;
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    bf16x8 o;
    o[0] = (__bf16)a.x, o[1] = (__bf16)a.y, o[2] = (__bf16)a.z, o[3] = (__bf16)a.w;
    o[4] = (__bf16)b.x, o[5] = (__bf16)b.y, o[6] = (__bf16)b.z, o[7] = (__bf16)b.w;
    const int64_t row = ((int64_t)n * Hp + py) * Wp + px;
    *reinterpret_cast<bf16x8 *>(y + row * ((int64_t)C * ps * ps) + ((int64_t)c * ps + ky) * ps + kx8 * 8) = o;
}

struct BnParams {
    const float *mean, *rstd, *w, *b;
};

// scale / shift of this thread's 8 channels: y = x * sc + sh
__device__ __forceinline__ void load_affine(const BnParams &p, int c0, float (&sc)[8], float (&sh)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float g = p.w ? p.w[c0 + j] : 1.f, r = p.rstd[c0 + j];
        sc[j] = r * g;
        sh[j] = (p.b ? p.b[c0 + j] : 0.f) - p.mean[c0 + j] * r * g;
    }
}

// MODE 0: [sum x | sum x^2];  MODE 1: [sum g' | sum g' xhat], g' = dy where relu'(BN(x)) != 0 (all of dy when !relu)
template <int MODE>
__global__ __launch_bounds__(256) void bn_nhwc_stats_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy,
                                                            int64_t rows, int C, BnParams p, int relu,
                                                            float *__restrict__ part) {
    __shared__ float s_red[256][17];
    const int C8 = C >> 3, cp = threadIdx.x % C8, rl = threadIdx.x / C8, RL = 256 / C8;
    const int c0 = cp * 8;
    float sc[8], sh[8], mu[8], rs[8];
    if (MODE == 1) {
        load_affine(p, c0, sc, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) mu[j] = p.mean[c0 + j], rs[j] = p.rstd[c0 + j];
    }
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = b[j] = 0.f;
    // contiguous slab of rows per workgroup
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x, r0 = per * blockIdx.x, r1 = r0 + per < rows ? r0 + per : rows;
    // 4 rows per step: the loads of a step are independent and in flight together
    auto add = [&](const bf16x8 &xv, const bf16x8 &gv) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (float)xv[j];
                a[j] += v;
                b[j] += v * v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (float)xv[j];
                const float g = (!relu || v * sc[j] + sh[j] > 0.f) ? (float)gv[j] : 0.f;
                a[j] += g;
                b[j] += g * (v - mu[j]) * rs[j];
            }
        }
    };
    int64_t r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {
        bf16x8 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = *reinterpret_cast<const bf16x8 *>(x + (r + u * RL) * C + c0);
            if (MODE == 1) gv[u] = *reinterpret_cast<const bf16x8 *>(dy + (r + u * RL) * C + c0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add(xv[u], gv[u]);
    }
    for (; r < r1; r += RL) {
        const bf16x8 xv = *reinterpret_cast<const bf16x8 *>(x + r * C + c0);
        bf16x8 gv = xv;
        if (MODE == 1) gv = *reinterpret_cast<const bf16x8 *>(dy + r * C + c0);
        add(xv, gv);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s_red[threadIdx.x][j] = a[j], s_red[threadIdx.x][8 + j] = b[j];
    __syncthreads();
    // thread t < 2C sums channel (t % C) of half (t / C) over the row lanes, in order
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        const int half = t / C, c = t - half * C;
        float s = 0.f;
        for (int l = 0; l < RL; ++l) s += s_red[l * C8 + (c >> 3)][half * 8 + (c & 7)];
        part[(int64_t)blockIdx.x * 2 * C + t] = s;
    }
}

// out[k] = sum over the partial rows, 32 columns x 8 part lanes per workgroup: lane l adds rows l, l + 8, ... in four
// independent chains (one serial chain over all rows is a latency-bound tail longer than the pass itself), the 8 lanes
// are added in order
__global__ __launch_bounds__(256) void bn_nhwc_sum_parts(const float *__restrict__ part, int nparts, int K, float *__restrict__ out) {
    __shared__ float s_p[8][33];
    const int col = threadIdx.x & 31, pl = threadIdx.x >> 5, k = blockIdx.x * 32 + col;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
        int i = pl;
        for (; i + 24 < nparts; i += 32) {
            s[0] += part[(int64_t)i * K + k];
            s[1] += part[(int64_t)(i + 8) * K + k];
            s[2] += part[(int64_t)(i + 16) * K + k];
            s[3] += part[(int64_t)(i + 24) * K + k];
        }
        for (; i < nparts; i += 8) s[0] += part[(int64_t)i * K + k];
    }
    s_p[pl][col] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (pl == 0 && k < K) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) t += s_p[l][col];
        out[k] = t;
    }
}

__global__ __launch_bounds__(256) void bn_nhwc_apply_kernel(const __bf16 *__restrict__ x, int64_t pieces, int C, BnParams p,
                                                            int relu, __bf16 *__restrict__ y) {
    const int C8 = C >> 3;
    const int64_t stride = (int64_t)gridDim.x * 256;              // a multiple of C8: the thread keeps its channels
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c0 = (int)(i % C8) * 8;
    float sc[8], sh[8];
    load_affine(p, c0, sc, sh);
    for (; i < pieces; i += stride) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = (float)v[j] * sc[j] + sh[j];
            o[j] = (__bf16)(relu ? fmaxf(t, 0.f) : t);
        }
        *reinterpret_cast<bf16x8 *>(y + i * 8) = o;
    }
}

// dx = w rstd (g' - mean(g') - xhat mean(g' xhat))
__global__ __launch_bounds__(256) void bn_nhwc_bwd_apply_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy,
                                                                int64_t pieces, int C, BnParams p, int relu,
                                                                const float *__restrict__ mg, const float *__restrict__ mgx,
                                                                __bf16 *__restrict__ dx) {
    const int C8 = C >> 3;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c0 = (int)(i % C8) * 8;
    float sc[8], sh[8], mu[8], rs[8], m0[8], m1[8];
    load_affine(p, c0, sc, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) mu[j] = p.mean[c0 + j], rs[j] = p.rstd[c0 + j], m0[j] = mg[c0 + j], m1[j] = mgx[c0 + j];
    for (; i < pieces; i += stride) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + i * 8);
        const bf16x8 gv = *reinterpret_cast<const bf16x8 *>(dy + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = (float)v[j];
            const float g = (!relu || xv * sc[j] + sh[j] > 0.f) ? (float)gv[j] : 0.f;
            o[j] = (__bf16)(sc[j] * (g - m0[j] - (xv - mu[j]) * rs[j] * m1[j]));
        }
        *reinterpret_cast<bf16x8 *>(dx + i * 8) = o;
    }
}

// MaxPool2d(3, 2, 1) on (N, H, W, C): output piece = 8 channels of one output pixel; idx = window position 0..8 of the
// first maximum in scan order (what torch's max_pool2d sends the gradient to)
__global__ __launch_bounds__(256) void maxpool_nhwc_fwd_kernel(const __bf16 *__restrict__ x, int N, int H, int W, int C, int OH,
                                                               int OW, __bf16 *__restrict__ y, uint8_t *__restrict__ idx) {
    const int C8 = C >> 3;
    const int64_t pieces = (int64_t)N * OH * OW * C8, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    const int c8 = (int)(i % C8);
    int64_t t = i / C8;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH), n = (int)(t / OH);
    float best[8];
    uint8_t bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = -INFINITY, bi[j] = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + (((int64_t)n * H + iy) * W + ix) * C + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j];
            if (f > best[j]) best[j] = f, bi[j] = (uint8_t)k;      // strict: ties keep the first position
        }
    }
    bf16x8 o;
    uint64_t packed = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)best[j], packed |= (uint64_t)bi[j] << (8 * j);
    *reinterpret_cast<bf16x8 *>(y + i * 8) = o;
    *reinterpret_cast<uint64_t *>(idx + i * 8) = packed;
}

__global__ __launch_bounds__(256) void maxpool_nhwc_bwd_kernel(const __bf16 *__restrict__ gy, const uint8_t *__restrict__ idx,
                                                               int N, int H, int W, int C, int OH, int OW,
                                                               __bf16 *__restrict__ gx) {
    const int C8 = C >> 3;
    const int64_t pieces = (int64_t)N * H * W * C8, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    const int c8 = (int)(i % C8);
    int64_t t = i / C8;
    const int ix = (int)(t % W);
    t /= W;
    const int iy = (int)(t % H), n = (int)(t / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // windows that contain (iy, ix): oy in [ceil((iy - 1) / 2), floor((iy + 1) / 2)]
    for (int oy = iy >> 1; oy <= (iy + 1) >> 1; ++oy) {
        if (oy >= OH) continue;
        for (int ox = ix >> 1; ox <= (ix + 1) >> 1; ++ox) {
            if (ox >= OW) continue;
            const int k = (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1));
            const int64_t o = (((int64_t)n * OH + oy) * OW + ox) * C + c8 * 8;
            const uint64_t packed = *reinterpret_cast<const uint64_t *>(idx + o);
            const bf16x8 g = *reinterpret_cast<const bf16x8 *>(gy + o);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if ((int)((packed >> (8 * j)) & 0xff) == k) acc[j] += (float)g[j];
        }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)acc[j];
    *reinterpret_cast<bf16x8 *>(gx + i * 8) = o;
}

int check_c(const char *fn, int64_t rows, int64_t C) {
    if (rows < 0 || C < 8 || C > 256 || (C & (C - 1))) return fail(VAH_E_SHAPE, "%s: C must be a power of two in [8, 256]", fn);
    return 0;
}

int stat_parts(int64_t rows) {       // at least 64 rows per partial
    const int64_t p = (rows + 63) / 64;
    return (int)(p < 1 ? 1 : (p > kStatParts ? kStatParts : p));
}

unsigned apply_grid(int64_t pieces) {
    const int64_t want = (pieces + 255) / 256;
    return (unsigned)(want < 8 * kCUs ? (want < 1 ? 1 : want) : 8 * kCUs);       // x 256 threads: a multiple of every C / 8
}

}  // namespace
}  // namespace vah

extern "C" {

int vah_image_to_nhwc16_bf16(const float *x, int64_t N, int64_t H, int64_t W, void *y, void *stream) {
    using namespace vah;
    clear_error();
    if (N < 0 || H < 1 || W < 1) return fail(VAH_E_SHAPE, "vah_image_to_nhwc16_bf16: bad dims");
    if (N == 0) return VAH_OK;
    if (!x || !y) return fail(VAH_E_NULL, "vah_image_to_nhwc16_bf16: null pointer");
    if ((uintptr_t)y % 16) return fail(VAH_E_ALIGN, "vah_image_to_nhwc16_bf16: y needs 16-byte alignment");
    const int64_t total = N * H * W;
    LaunchScope scope("spm_image_to_nhwc", total * (12 + 32), (hipStream_t)stream);
    hipLaunchKernelGGL(image_to_nhwc16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, H * W,
                       total, (__bf16 *)y);
    return check_launch("image_to_nhwc16");
}

int64_t vah_bn_nhwc_ws_floats(int64_t C) { return vah::kStatParts * 2 * C; }

/* sums[2C] = [sum x | sum x^2] over the rows of x (rows, C) bf16 */
int vah_bn_nhwc_stats(const void *x, int64_t rows, int64_t C, float *sums, float *ws, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_nhwc_stats";
    if (int rc = check_c(fn, rows, C)) return rc;
    if (!x || !sums || !ws) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)stream;
    const int parts = stat_parts(rows);
    LaunchScope scope("spm_bn_stats", rows * C * 2, st);
    hipLaunchKernelGGL(bn_nhwc_stats_kernel<0>, dim3(parts), dim3(256), 0, st, (const __bf16 *)x, (const __bf16 *)nullptr, rows,
                       (int)C, BnParams{}, 0, ws);
    if (int rc = check_launch(fn)) return rc;
    hipLaunchKernelGGL(bn_nhwc_sum_parts, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), 0, st, (const float *)ws, parts,
                       (int)(2 * C), sums);
    return check_launch(fn);
}

int vah_bn_nhwc_apply(const void *x, int64_t rows, int64_t C, const float *mean, const float *rstd, const float *w,
                      const float *b, int relu, void *y, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_nhwc_apply";
    if (int rc = check_c(fn, rows, C)) return rc;
    if (rows == 0) return VAH_OK;
    if (!x || !y || !mean || !rstd) return fail(VAH_E_NULL, "%s: null pointer", fn);
    const int64_t pieces = rows * C / 8;
    LaunchScope scope("spm_bn_apply", rows * C * 4, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_nhwc_apply_kernel, dim3(apply_grid(pieces)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, pieces,
                       (int)C, BnParams{mean, rstd, w, b}, relu, (__bf16 *)y);
    return check_launch(fn);
}

/* sums[2C] = [sum g' | sum g' xhat] */
int vah_bn_nhwc_bwd_stats(const void *x, const void *dy, int64_t rows, int64_t C, const float *mean, const float *rstd,
                          const float *w, const float *b, int relu, float *sums, float *ws, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_nhwc_bwd_stats";
    if (int rc = check_c(fn, rows, C)) return rc;
    if (!x || !dy || !sums || !ws || !mean || !rstd) return fail(VAH_E_NULL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)stream;
    const int parts = stat_parts(rows);
    LaunchScope scope("spm_bn_bwd_stats", rows * C * 4, st);
    hipLaunchKernelGGL(bn_nhwc_stats_kernel<1>, dim3(parts), dim3(256), 0, st, (const __bf16 *)x, (const __bf16 *)dy, rows, (int)C,
                       BnParams{mean, rstd, w, b}, relu, ws);
    if (int rc = check_launch(fn)) return rc;
    hipLaunchKernelGGL(bn_nhwc_sum_parts, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), 0, st, (const float *)ws, parts,
                       (int)(2 * C), sums);
    return check_launch(fn);
}

int vah_bn_nhwc_bwd_apply(const void *x, const void *dy, int64_t rows, int64_t C, const float *mean, const float *rstd,
                          const float *w, const float *b, int relu, const float *mean_g, const float *mean_gx, void *dx,
                          void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_bn_nhwc_bwd_apply";
    if (int rc = check_c(fn, rows, C)) return rc;
    if (rows == 0) return VAH_OK;
    if (!x || !dy || !dx || !mean || !rstd || !mean_g || !mean_gx) return fail(VAH_E_NULL, "%s: null pointer", fn);
    const int64_t pieces = rows * C / 8;
    LaunchScope scope("spm_bn_bwd_apply", rows * C * 6, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_nhwc_bwd_apply_kernel, dim3(apply_grid(pieces)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x,
                       (const __bf16 *)dy, pieces, (int)C, BnParams{mean, rstd, w, b}, relu, mean_g, mean_gx, (__bf16 *)dx);
    return check_launch(fn);
}

int vah_maxpool3s2_nhwc_fwd_bf16(const void *x, int64_t N, int64_t H, int64_t W, int64_t C, void *y, void *idx, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_maxpool3s2_nhwc_fwd_bf16";
    if (N < 0 || H < 1 || W < 1 || C < 8 || C % 8 || H > 32767 || W > 32767) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (N == 0) return VAH_OK;
    if (!x || !y || !idx) return fail(VAH_E_NULL, "%s: null pointer", fn);
    const int64_t OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1, pieces = N * OH * OW * (C / 8);
    LaunchScope scope("spm_maxpool_fwd", N * H * W * C * 2 + N * OH * OW * C * 3, (hipStream_t)stream);
    hipLaunchKernelGGL(maxpool_nhwc_fwd_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)x, (int)N, (int)H, (int)W, (int)C, (int)OH, (int)OW, (__bf16 *)y, (uint8_t *)idx);
    return check_launch(fn);
}

int vah_maxpool3s2_nhwc_bwd_bf16(const void *gy, const void *idx, int64_t N, int64_t H, int64_t W, int64_t C, void *gx,
                                 void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_maxpool3s2_nhwc_bwd_bf16";
    if (N < 0 || H < 1 || W < 1 || C < 8 || C % 8 || H > 32767 || W > 32767) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (N == 0) return VAH_OK;
    if (!gy || !gx || !idx) return fail(VAH_E_NULL, "%s: null pointer", fn);
    const int64_t OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1, pieces = N * H * W * (C / 8);
    LaunchScope scope("spm_maxpool_bwd", N * H * W * C * 2 + N * OH * OW * C * 3, (hipStream_t)stream);
    hipLaunchKernelGGL(maxpool_nhwc_bwd_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)gy, (const uint8_t *)idx, (int)N, (int)H, (int)W, (int)C, (int)OH, (int)OW, (__bf16 *)gx);
    return check_launch(fn);
}


int vah_patchify_bf16(const float *x, int64_t N, int64_t C, int64_t H, int64_t W, int64_t ps, void *y, void *stream) {
    using namespace vah;
    clear_error();
    const char *fn = "vah_patchify_bf16";
    if (N < 0 || C < 1 || ps < 8 || ps % 8 || H < ps || W < ps || H % ps || W % ps) return fail(VAH_E_SHAPE, "%s: bad dims", fn);
    if (N == 0) return VAH_OK;
    if (!x || !y) return fail(VAH_E_NULL, "%s: null pointer", fn);
    if (((uintptr_t)x | (uintptr_t)y) % 16) return fail(VAH_E_ALIGN, "%s: 16-byte alignment", fn);
    const int64_t items = N * C * H * W / 8;
    LaunchScope scope("patchify", N * C * H * W * 6, (hipStream_t)stream);
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (int)C, (int)H,
                       (int)W, (int)ps, items, (__bf16 *)y);
    return check_launch(fn);
}

}  // extern "C"
