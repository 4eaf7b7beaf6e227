// to_fixed64 in isolation: random values, rare tiny ones (divergent negative-shift branch), LDS u64 atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <random>
__device__ __forceinline__ long long to_fixed64(float v, int k) {
    int ex;
    const float f = frexpf(v, &ex);
    const long long m = (long long)(int)ldexpf(f, 24);
    const int sh = k + ex - 24;
    if (sh >= 0) return m << sh;
    if (sh > -26) return (m + (1ll << (-sh - 1))) >> (-sh);
    return 0;
}
__device__ __forceinline__ long long to_fixed64_dbl(float v, int k) {  // round-half-even
    return __double2ll_rn(ldexp((double)v, k));
}
__device__ __forceinline__ long long to_fixed64_nobranch(float v, int k) {
    int ex;
    const float f = frexpf(v, &ex);
    const long long m = (long long)(int)ldexpf(f, 24);
    const int sh = k + ex - 24;
    const int l = sh > 0 ? sh : 0;
    int r = sh < 0 ? -sh : 0;
    r = r > 40 ? 40 : r;
    const long long half = (1ll << r) >> 1;
    return ((m << l) + half) >> r;
}
__device__ __forceinline__ long long to_fixed64_split(float v, int k) {  // fp32 + 32-bit ops only, round-half-even
    const float a = fabsf(v);
    const float t = ldexpf(a, k - 32);
    const float hf = floorf(t);
    const uint32_t hi = (uint32_t)(int)hf;
    const uint32_t lo = (uint32_t)rintf(ldexpf(t - hf, 32));
    const long long q = (long long)(((unsigned long long)hi << 32) | lo);
    return v < 0 ? -q : q;
}
template <int VAR> __device__ __forceinline__ long long cvt(float v, int k) {
    if (VAR == 4) return to_fixed64_split(v, k);
    if (VAR == 0) return to_fixed64(v, k);
    if (VAR == 1) return to_fixed64_dbl(v, k);
    if (VAR == 2) return to_fixed64_nobranch(v, k);
    long long r = to_fixed64(v, k);
    asm volatile("s_nop 7" ::: "memory");
    return r;
}
static long long h_fixed(float v, int k) {
    int ex; float f = frexpf(v, &ex);
    long long m = (long long)(int)ldexpf(f, 24);
    int sh = k + ex - 24;
    if (sh >= 0) return m << sh;
    if (sh > -26) return (m + (1ll << (-sh - 1))) >> (-sh);
    return 0;
}
__global__ void k_conv(const float2* v, long long* out, int n, int k) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = to_fixed64(v[i].x, k); out[2 * i + 1] = to_fixed64(v[i].y, k); }
}
template <int VAR> __global__ void __launch_bounds__(1024) k_acc(const uint32_t* keys, const float2* vals, int n, int k, unsigned long long* out, int rows) {
    extern __shared__ unsigned long long acc[];
    for (int i = threadIdx.x; i < rows * 2; i += 1024) acc[i] = 0;
    __syncthreads();
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        uint32_t kk[4]; float2 vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { kk[u] = keys[i + u * 1024]; vv[u] = vals[i + u * 1024]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            atomicAdd(&acc[kk[u] * 2], (unsigned long long)cvt<VAR>(vv[u].x, k));
            atomicAdd(&acc[kk[u] * 2 + 1], (unsigned long long)cvt<VAR>(vv[u].y, k));
        }
    }
    for (; i < n; i += 1024) {
        atomicAdd(&acc[keys[i] * 2], (unsigned long long)cvt<VAR>(vals[i].x, k));
        atomicAdd(&acc[keys[i] * 2 + 1], (unsigned long long)cvt<VAR>(vals[i].y, k));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < rows * 2; j += 1024) out[j] = acc[j];
}
int main() {
    const int n = 1 << 20, rows = 8192, k = 50;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1e-3f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<float2> v(n); std::vector<uint32_t> keys(n);
    for (int i = 0; i < n; i++) {
        float w0 = ud(rng) < 0.02f ? ud(rng) * 1e-5f : ud(rng);
        float w1 = ud(rng) < 0.02f ? ud(rng) * 1e-6f : ud(rng);
        v[i] = make_float2(nd(rng) * w0, nd(rng) * w1);
        keys[i] = rng() % rows;
    }
    float2* dv; long long* dout; uint32_t* dk; unsigned long long* dacc;
    hipMalloc(&dv, n * 8); hipMalloc(&dout, n * 16); hipMalloc(&dk, n * 4); hipMalloc(&dacc, rows * 16);
    hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dk, keys.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<long long> out(2 * n);
    int bad = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(k_conv, dim3(n / 256), dim3(256), 0, 0, dv, dout, n, k);
        hipMemcpy(out.data(), dout, n * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) {
            if (out[2 * i] != h_fixed(v[i].x, k) || out[2 * i + 1] != h_fixed(v[i].y, k)) {
                if (bad < 5) printf("conv mismatch i=%d v=(%g,%g) got (%lld,%lld) want (%lld,%lld)\n", i, v[i].x, v[i].y, out[2*i], out[2*i+1], h_fixed(v[i].x, k), h_fixed(v[i].y, k));
                bad++;
            }
        }
    }
    printf("conv mismatches: %d\n", bad);
    std::vector<long long> want(rows * 2, 0), got(rows * 2);
    for (int i = 0; i < n; i++) { want[keys[i] * 2] += h_fixed(v[i].x, k); want[keys[i] * 2 + 1] += h_fixed(v[i].y, k); }
    auto run = [&](auto kern, const char* name, bool half_even) {
        std::vector<long long> w2(rows * 2, 0);
        for (int i = 0; i < n; i++) {
            auto hf = [&](float x) { return half_even ? (long long)llrint(ldexp((double)x, k)) : h_fixed(x, k); };
            w2[keys[i] * 2] += hf(v[i].x); w2[keys[i] * 2 + 1] += hf(v[i].y);
        }
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, rows * 16);
        int abad = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(1024), rows * 16, 0, dk, dv, n, k, dacc, rows);
            hipMemcpy(got.data(), dacc, rows * 16, hipMemcpyDeviceToHost);
            for (int j = 0; j < rows * 2; j++) if (got[j] != w2[j]) { if (abad < 2) printf("%s mismatch rep %d j=%d diff %lld\n", name, rep, j, got[j] - w2[j]); abad++; }
        }
        printf("%s: acc mismatches: %d\n", name, abad);
    };
    run(k_acc<0>, "branchy", false);
    run(k_acc<1>, "double", true);
    run(k_acc<2>, "nobranch", false);
    run(k_acc<3>, "branchy+nop", false);
    run(k_acc<4>, "fp32split", true);
    for (int kk : {50, 40, 61, 33, 20}) {   // other scales: values are ~1e-3 * w, so |v| * 2^kk stays below 2^62
        std::vector<long long> w2(rows * 2, 0);
        for (int i = 0; i < n; i++) { w2[keys[i] * 2] += (long long)llrint(ldexp((double)v[i].x, kk)); w2[keys[i] * 2 + 1] += (long long)llrint(ldexp((double)v[i].y, kk)); }
        hipLaunchKernelGGL(k_acc<4>, dim3(1), dim3(1024), rows * 16, 0, dk, dv, n, kk, dacc, rows);
        hipMemcpy(got.data(), dacc, rows * 16, hipMemcpyDeviceToHost);
        int bad2 = 0;
        for (int j = 0; j < rows * 2; j++) bad2 += got[j] != w2[j];
        printf("fp32split k=%d: mismatches %d\n", kk, bad2);
    }
    return 0;
}
