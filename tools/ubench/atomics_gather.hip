// Micro-benchmark (GPU box): random 4/8-byte gathers and fp atomics over tables of various sizes,
// with and without XCD-partitioned addressing.  Guides the gridencoder kernel design.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: every block addresses the whole table; mode 1: block b only addresses slice (b%8) of 8
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* tab, uint32_t rows, uint32_t per_thread, int part, float* sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = rng(tid * 2654435761u + 12345u);
    float acc = 0;
    const uint32_t slice = rows / 8, base = part ? (blockIdx.x % 8) * slice : 0, span = part ? slice : rows;
#pragma unroll 8
    for (uint32_t i = 0; i < per_thread; i++) {
        s = rng(s + i);
        const uint32_t r = base + (s % span);
        if (OP == 0) acc += __uint_as_float(tab[r]);                                  // 4-byte gather
        else if (OP == 1) { uint2 v = reinterpret_cast<uint2*>(tab)[r >> 1]; acc += __uint_as_float(v.x ^ v.y); }  // 8-byte
        else if (OP == 2) unsafeAtomicAdd(reinterpret_cast<float*>(tab) + r, 1.0f);
        else if (OP == 3) unsafeAtomicAdd(reinterpret_cast<__half2*>(tab) + r, __halves2half2(__float2half(1.f), __float2half(1.f)));
        else if (OP == 4) atomicAdd(tab + r, 1u);
        else if (OP == 5) { uint4 v = reinterpret_cast<uint4*>(tab)[r >> 2]; acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w); }  // 16-byte
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_lds(float* out, uint32_t per_thread) {
    __shared__ float lds[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t s = rng(blockIdx.x * 256 + threadIdx.x);
    for (uint32_t i = 0; i < per_thread; i++) { s = rng(s + i); atomicAdd(&lds[s & 32767], 1.0f); }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}

template <int OP>
void run(const char* name, uint32_t* tab, uint32_t rows, int part, float* sink) {
    const uint32_t blocks = 2048, per = 512;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, tab, rows, per, part, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, tab, rows, per, part, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double ops = (double)blocks * 256 * per;
    printf("%-14s rows=%9u (%6.1f MB) part=%d : %8.1f us  %7.1f Gops/s\n", name, rows, rows * 4 / 1e6, part, ms * 1e3, ops / ms / 1e6);
}

int main() {
    float* sink; CK(hipMalloc(&sink, 4096));
    const uint32_t sizes[] = {4920u, 1u << 19, 1u << 21, 1u << 23, 1u << 25};
    uint32_t* tab; CK(hipMalloc(&tab, (size_t)(1u << 25) * 4)); CK(hipMemset(tab, 0, (size_t)(1u << 25) * 4));
    for (uint32_t rows : sizes) for (int part = 0; part < 2; part++) {
        if (rows < 64 && part) continue;
        run<0>("gather4", tab, rows, part, sink);
        run<1>("gather8", tab, rows, part, sink);
        run<5>("gather16", tab, rows, part, sink);
        run<2>("atomic_f32", tab, rows, part, sink);
        run<3>("atomic_pkf16", tab, rows, part, sink);
        run<4>("atomic_u32", tab, rows, part, sink);
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_lds, dim3(2048), dim3(256), 0, 0, sink, 2048u);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_lds, dim3(2048), dim3(256), 0, 0, sink, 2048u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("lds_atomic_f32 128KB: %.1f us  %.1f Gops/s\n", ms * 1e3, 2048.0 * 256 * 2048 / ms / 1e6);
    return 0;
}
