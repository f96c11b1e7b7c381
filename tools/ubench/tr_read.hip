// What does ds_read_b64_tr_b16 hand to each lane?  LDS holds G[row][col] = row * 32 + col as 16-bit ints
// (64 rows x 32 columns, 64-byte rows); every lane issues the address msda_tile.hip uses for k-step s = 0,
// first read (r = 0), and prints the 4 values it receives as (row, col) pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__global__ void k(short *out) {
    __shared__ __attribute__((aligned(16))) short G[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) G[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    const int grp = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const short *gbase = G + (8 * (grp >> 1) + qq) * 32 + 16 * (grp & 1) + 4 * pp;
    s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(gbase));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = t0[j];
}
int main() {
    short *d, h[256];
    hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d (want col %2d rows %d..%d):", l, l & 31, 8 * (l >> 5), 8 * (l >> 5) + 3);
        for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / 32, h[l * 4 + j] % 32);
        printf("\n");
    }
    return 0;
}
