// Micro-benchmark (GPU box): what bounds a workgroup that streams the whole point array (all workgroups read the same data)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// V: 0 = 3 dword loads (stride 12), 1 = one float4 load (16 B/pt), UNROLL points in flight, STAG: per-block start offset
template <int V, int UNROLL, int STAG, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const float* __restrict__ pts, const float4* __restrict__ pts4, uint32_t B, float* out) {
    float junk = 0;
    const uint32_t per_round = THREADS * UNROLL, rounds = B / per_round;
    const uint32_t start = STAG ? (blockIdx.x * 7919u) % rounds : 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint32_t rr = start + r; if (rr >= rounds) rr -= rounds;
        const uint32_t b0 = rr * per_round + threadIdx.x;
        float x[UNROLL], y[UNROLL], z[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t b = b0 + u * THREADS;
            if (V == 0) { x[u] = pts[b * 3]; y[u] = pts[b * 3 + 1]; z[u] = pts[b * 3 + 2]; }
            else { const float4 p = pts4[b]; x[u] = p.x; y[u] = p.y; z[u] = p.z; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) junk += x[u] * 1.0001f + y[u] + z[u];
    }
    if (junk == 12345.678f) out[0] = junk;
}

template <int V, int UNROLL, int STAG, int THREADS>
void run(const char* name, const float* pts, const float4* pts4, uint32_t B, uint32_t nblocks, float* out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<V, UNROLL, STAG, THREADS>), dim3(nblocks), dim3(THREADS), 0, 0, pts, pts4, B, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<V, UNROLL, STAG, THREADS>), dim3(nblocks), dim3(THREADS), 0, 0, pts, pts4, B, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-40s blocks=%4u thr=%4d : %8.1f us  %.3f ns/block-point  %.0f GB/s/CU-equivalent\n", name, nblocks, THREADS, ms * 1e3,
           ms * 1e6 / ((double)nblocks * B) * 256, (V ? 16.0 : 12.0) * nblocks * B / 256 / (ms * 1e6));
}

int main() {
    const uint32_t B = 131072;
    float* h = (float*)malloc(B * 16);
    for (uint32_t i = 0; i < B * 4; i++) h[i] = (float)(rand() & 0xffff) / 65536.0f;
    float* pts; CK(hipMalloc(&pts, B * 12)); CK(hipMemcpy(pts, h, B * 12, hipMemcpyHostToDevice));
    float4* pts4; CK(hipMalloc(&pts4, B * 16)); CK(hipMemcpy(pts4, h, B * 16, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, 4096));
    run<0, 1, 0, 1024>("3xdword u1", pts, pts4, B, 512, out);
    run<0, 4, 0, 1024>("3xdword u4", pts, pts4, B, 512, out);
    run<0, 4, 1, 1024>("3xdword u4 staggered", pts, pts4, B, 512, out);
    run<0, 16, 1, 1024>("3xdword u16 staggered", pts, pts4, B, 512, out);
    run<1, 1, 0, 1024>("float4 u1", pts, pts4, B, 512, out);
    run<1, 4, 1, 1024>("float4 u4 staggered", pts, pts4, B, 512, out);
    run<1, 8, 1, 1024>("float4 u8 staggered", pts, pts4, B, 512, out);
    run<1, 8, 1, 256>("float4 u8 staggered thr256", pts, pts4, B, 2048, out);
    run<1, 8, 1, 1024>("float4 u8 staggered 256 blocks", pts, pts4, B, 256, out);
    return 0;
}
