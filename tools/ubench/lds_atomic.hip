// Microbenchmark: LDS float atomic add rate on gfx950 (design input for the MSDA backward tile kernels).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic.hip -o gpurun_out/lds_atomic && gpurun_out/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: ds_add_f32, each half-wave adds one whole 32-float row (random row per step)
// MODE 1: same with plain ds_write_b32 (store) for comparison
// MODE 2: ds_add_f32, 8 lanes x 4 consecutive floats per row (4 instructions), 8 rows per wave
// MODE 3: ds_read_b128 row gather (8 lanes x float4), for the read side
// MODE 4: ds_pk_add_bf16 (one dword = 2 bf16 per lane), 16 lanes per 32-channel row, 4 rows per wave
template <int MODE>
__global__ __launch_bounds__(256) void k(const int *__restrict__ rows, int nrows_lds, int iters, float *out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < nrows_lds * 32; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    const int *rp = rows + (blockIdx.x * 4 + wave) * iters * 8;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            const int r = rp[it * 8 + (lane >> 5)];
            atomicAdd(&lds[r * 32 + (lane & 31)], 1.0f);
        } else if (MODE == 1) {
            const int r = rp[it * 8 + (lane >> 5)];
            lds[r * 32 + (lane & 31)] = (float)it;
        } else if (MODE == 2) {
            const int r = rp[it * 8 + (lane >> 3)];
            float *d = &lds[r * 32 + (lane & 7) * 4];
            atomicAdd(d + 0, 1.0f);
            atomicAdd(d + 1, 1.0f);
            atomicAdd(d + 2, 1.0f);
            atomicAdd(d + 3, 1.0f);
        } else if (MODE == 3) {
            const int r = rp[it * 8 + (lane >> 3)];
            const float4 v = *reinterpret_cast<const float4 *>(&lds[r * 32 + (lane & 7) * 4]);
            acc += v.x + v.y + v.z + v.w;
        } else if (MODE == 4) {
            const int r = rp[it * 8 + (lane >> 4)];
            typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf2;
            bf2 v; v[0] = (__bf16)1.0f; v[1] = (__bf16)1.0f;
            unsigned addr = (unsigned)(size_t)(&lds[r * 16 + (lane & 15)]);
            asm volatile("ds_pk_add_bf16 %0, %1" :: "v"(addr), "v"(v) : "memory");
        }
    }
    __syncthreads();
    float s = acc;
    for (int i = threadIdx.x; i < nrows_lds * 32; i += 256) s += lds[i];
    if (s == -1.f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = lds[rows[0] * 32];
}

template <int MODE>
void run(const char *name, int wg_per_cu, int nrows_lds, int iters, const int *d_rows, float *d_out, double lanes_per_iter) {
    const int grid = 256 * wg_per_cu;
    const size_t smem = (size_t)nrows_lds * 128;
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<grid, 256, smem>>>(d_rows, nrows_lds, iters, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) k<MODE><<<grid, 256, smem>>>(d_rows, nrows_lds, iters, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double lane_ops = (double)grid * 4 * iters * lanes_per_iter;
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-44s wg/cu %d rows %4d: %8.1f us  %.2f lane-ops/clk/CU (at 2.4 GHz)  %.1f G lane-ops/s chip\n", name, wg_per_cu,
           nrows_lds, ms * 1e3, lane_ops / 256 / clk, lane_ops / (ms * 1e-3) / 1e9);
}

int main() {
    const int iters = 4096;
    const int maxwg = 256 * 8;
    std::vector<int> rows((size_t)maxwg * 4 * iters * 8);
    float *d_out; int *d_rows;
    CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_rows, rows.size() * 4));
    for (int nrows : {64, 324, 1024}) {
        srand(1);
        for (auto &r : rows) r = rand() % nrows;
        CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
        for (int wpc : {1, 2, 4, 8}) {
            if ((size_t)nrows * 128 * wpc > 160 * 1024) continue;
            run<0>("ds_add_f32 32-lane rows", wpc, nrows, iters, d_rows, d_out, 64);
            run<1>("ds_write_b32 32-lane rows", wpc, nrows, iters, d_rows, d_out, 64);
            run<2>("ds_add_f32 8 lanes x 4 floats (4 instr)", wpc, nrows, iters, d_rows, d_out, 256);
            run<3>("ds_read_b128 8 lanes x float4", wpc, nrows, iters, d_rows, d_out, 256);
            run<4>("ds_pk_add_bf16 16 lanes x 2 (rows of 64 B)", wpc, nrows, iters, d_rows, d_out, 128);
        }
    }
    return 0;
}
