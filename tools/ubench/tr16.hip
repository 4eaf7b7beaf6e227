// ds_read_b64_tr_b16 (gfx950) semantics probe: LDS image M[row][col] (row stride given in halfs) holds the value 100*row + col;
// every lane of a 16-lane group supplies the address of row (i / 4), cols 4 (i % 4) .. +3 of its group's block; prints what
// each lane receives.  Expected (if the read transposes the group's 4 x 16 block): lane i gets column i, rows 0..3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out, int stride) {
    __shared__ __attribute__((aligned(16))) _Float16 M[64 * 80];
    for (int i = threadIdx.x; i < 64 * 80; i += 64) M[i] = (_Float16)0;
    __syncthreads();
    for (int r = 0; r < 32; r++)
        for (int c = threadIdx.x; c < 64; c += 64) M[r * stride + c] = (_Float16)(float)(64 * r + c);  // < 2048: exact
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    // group g: block rows 4 * (g >> 1) .., cols 16 * (g & 1) ..
    const _Float16* p = M + (4 * (g >> 1) + i / 4) * stride + 16 * (g & 1) + 4 * (i % 4);
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p);
    for (int j = 0; j < 4; j++) out[lane * 4 + j] = (float)v[j];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    for (int stride : {64, 72}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d\n", stride);
        for (int l = 0; l < 64; l++) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; j++) printf(" (r%d,c%d)", (int)h[l * 4 + j] / 64, (int)h[l * 4 + j] % 64);
            printf("\n");
        }
    }
    return 0;
}
