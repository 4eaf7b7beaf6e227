// Micro-benchmark (GPU box): LDS atomic rates by type, random addresses in a 128 KiB LDS array.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t xs(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, uint32_t per_thread) {
    __shared__ uint32_t lds[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 747796405u + 2891336453u;
#pragma unroll 8
    for (uint32_t i = 0; i < per_thread; i++) {
        s = xs(s);
        if (OP == 0) atomicAdd(&lds[s & 32767], 1u);
        else if (OP == 1) atomicAdd(reinterpret_cast<float*>(lds) + (s & 32767), 1.0f);
        else if (OP == 2) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + (s & 16383), 1ull);
        else if (OP == 3) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 v = {(_Float16)1.0f, (_Float16)1.0f};
                            __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)(lds + (s & 32767)), v); }
        else if (OP == 4) lds[s & 32767] = s;                       // plain store
        else if (OP == 5) out[0] += __uint_as_float(lds[s & 32767]);  // plain load (never true sink)
        else if (OP == 6) atomicMax(&lds[s & 32767], s);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x + 1] = __uint_as_float(lds[5]);
}
template <int OP> void run(const char* name, float* out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, out, 1024u);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, out, 1024u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-16s %8.1f us  %8.1f Gops/s\n", name, ms * 1e3, 2048.0 * 256 * 1024 / ms / 1e6);
}
int main() {
    float* out; CK(hipMalloc(&out, 65536));
    run<0>("lds_add_u32", out); run<1>("lds_add_f32", out); run<2>("lds_add_u64", out); run<3>("lds_pk_add_f16", out);
    run<4>("lds_store_b32", out); run<6>("lds_max_u32", out);
    return 0;
}
