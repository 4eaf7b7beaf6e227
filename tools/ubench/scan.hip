// Micro-benchmark (GPU box): cost anatomy of the "every workgroup scans all points" backward sweep.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int V, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const float* __restrict__ pts, uint32_t B, float scale, uint32_t nslices, unsigned long long* out) {
    extern __shared__ unsigned long long acc[];
    const uint32_t slice = blockIdx.x % nslices;
    const uint32_t mask = (1u << 19) - 1, slice_rows = (1u << 19) / nslices, row0 = slice * slice_rows;
    for (uint32_t i = threadIdx.x; i < slice_rows * 2; i += THREADS) acc[i] = 0;
    __syncthreads();
    float junk = 0;
    for (uint32_t b = threadIdx.x; b < B; b += THREADS) {
        const float x = pts[b * 3], y = pts[b * 3 + 1], z = pts[b * 3 + 2];
        if (V == 0) { junk += x + y + z; continue; }
        const float px = fmaf(x, scale, 0.5f), py = fmaf(y, scale, 0.5f), pz = fmaf(z, scale, 0.5f);
        const uint32_t gx = (uint32_t)floorf(px), gy = (uint32_t)floorf(py), gz = (uint32_t)floorf(pz);
        const float fx = px - gx, fy = py - gy, fz = pz - gz;
        const uint32_t hy0 = gy * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = gz * 805459861u, hz1 = hz0 + 805459861u;
        uint32_t hits = 0;
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            const uint32_t h = ((idx & 1) ? gx + 1 : gx) ^ ((idx & 2) ? hy1 : hy0) ^ ((idx & 4) ? hz1 : hz0);
            hits |= (((h & mask) - row0) < slice_rows ? 1u : 0u) << idx;
        }
        if (V == 1) { junk += hits + fx + fy + fz; continue; }
        while (hits) {
            const uint32_t idx = __builtin_ctz(hits);
            hits &= hits - 1;
            const uint32_t h = ((idx & 1) ? gx + 1 : gx) ^ ((idx & 2) ? hy1 : hy0) ^ ((idx & 4) ? hz1 : hz0);
            const float w = ((idx & 1) ? fx : 1 - fx) * ((idx & 2) ? fy : 1 - fy) * ((idx & 4) ? fz : 1 - fz);
            const uint32_t local = (h & mask) - row0;
            atomicAdd(&acc[local * 2], (unsigned long long)(long long)(w * 1048576.0f));
            atomicAdd(&acc[local * 2 + 1], (unsigned long long)(long long)(w * 524288.0f));
        }
    }
    __syncthreads();
    if (junk == 12345.678f) out[0] = 1;
    if (threadIdx.x == 0) out[blockIdx.x + 1] = acc[3];
}

template <int V, int THREADS>
void run(const char* name, const float* pts, uint32_t B, uint32_t nblocks, uint32_t nslices, unsigned long long* out) {
    const size_t lds = (size_t)((1u << 19) / nslices) * 16;
    CK(hipFuncSetAttribute((const void*)&k<V, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<V, THREADS>), dim3(nblocks), dim3(THREADS), lds, 0, pts, B, 2047.0f, nslices, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<V, THREADS>), dim3(nblocks), dim3(THREADS), lds, 0, pts, B, 2047.0f, nslices, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-34s blocks=%4u threads=%4d lds=%6zu : %8.1f us   %.2f ns per block-point\n", name, nblocks, THREADS, lds, ms * 1e3,
           ms * 1e6 / ((double)nblocks * B) * 256);
}

int main() {
    const uint32_t B = 131072;
    float* h = (float*)malloc(B * 12);
    for (uint32_t i = 0; i < B * 3; i++) h[i] = (float)rand() / RAND_MAX;
    float* pts; CK(hipMalloc(&pts, B * 12)); CK(hipMemcpy(pts, h, B * 12, hipMemcpyHostToDevice));
    unsigned long long* out; CK(hipMalloc(&out, 1 << 20));
    run<0, 1024>("V0 load only", pts, B, 512, 52, out);
    run<1, 1024>("V1 +index/hits", pts, B, 512, 52, out);
    run<2, 1024>("V2 +lds atomics", pts, B, 512, 52, out);
    run<0, 256>("V0 load only", pts, B, 2048, 208, out);
    run<1, 256>("V1 +index/hits", pts, B, 2048, 208, out);
    run<2, 256>("V2 +lds atomics", pts, B, 2048, 208, out);
    run<1, 512>("V1 +index/hits", pts, B, 1024, 104, out);
    run<2, 512>("V2 +lds atomics", pts, B, 1024, 104, out);
    run<2, 1024>("V2 256 blocks (1 round)", pts, B, 256, 52, out);
    return 0;
}
