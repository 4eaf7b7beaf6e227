// gridencoder.hip — multiresolution hash / tiled grid encoding for gfx950.
// Replaces gridencoder/src/gridencoder.cu of the reference (forward :87-242, backward :245-366,
// total-variation gradient :503-607) behind the C ABI of include/seal3d_hip.h.
//
// MI355X mapping
//  * XCD-aware level placement.  MI355X has 8 XCDs with private 4 MiB L2s and workgroup b is
//    observed to run on XCD b % 8.  Workgroup b serves levels {l : l % 8 == b % 8} for point chunk
//    b / 8, so one XCD's L2 only ever holds ITS levels' tables (two tables for L = 16: one coarse,
//    one 2 MiB hashed) instead of all sixteen.  Placement affects speed only — any block→XCD map
//    gives the same result.
//  * One lane = one point for all levels of its XCD: 8 corners x (L/8) levels of independent
//    gathers in flight per lane, features of a corner fetched with ONE load (C*sizeof(T) bytes).
//  * Level-major output [L,B,C] (the reference layout): lanes write consecutive addresses.
//  * Arithmetic follows the oracle expression by expression (explicit fmaf; fp16 tables use
//    half accumulators with one rounding per op like the reference's at::Half registers), so
//    corner rows are bit-exact and outputs match the oracle to the last bit.
//  * The per-level scale table is computed once on the host (glibc exp2f) and passed by value.
//  * Backward.  Measured on MI355X: global fp atomics retire at a flat ~21 G/s chip-wide (memory-side,
//    independent of table size or XCD locality) and LDS *float* atomics at ~0.2 T/s, while LDS *integer*
//    atomics run at ~2.3 T/s (tools/ubench).  The table gradient is therefore accumulated in LDS as 64-bit
//    fixed point: the table is cut into 160 KiB slices, one workgroup owns one (level, slice), scans all
//    points, adds the contributions that land in its slice with ds_add_u64, and writes the slice back with
//    coalesced stores.  Integer adds commute, so the result is bit-reproducible run to run (the reference's
//    atomics are not) and carries ~40 bits below the largest |grad| of the level — more accurate than fp32
//    atomics.  Small batches keep the direct-atomic kernel (fixed cost of the slice sweep ~25 us).
#include "s3d_common.hpp"
#include <math.h>
#include <type_traits>

namespace s3d {
namespace {

constexpr uint32_t kMaxLevels = 32;
struct LevelScales { float v[kMaxLevels]; };

constexpr uint32_t kPrimes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};

template <uint32_t D>
__device__ __forceinline__ uint32_t grid_row(uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                             uint32_t resolution, const uint32_t (&pg)[D]) {
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pg[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t r = 0;
#pragma unroll
        for (uint32_t i = 0; i < D; i++) r ^= pg[i] * kPrimes[i];
        index = r;
    }
    return index % hashmap_size;
}

// ---- feature vector load/store: one memory instruction per corner ----
template <typename T, uint32_t C> struct FeatVec;
template <> struct FeatVec<float, 1> { using type = float; };
template <> struct FeatVec<float, 2> { using type = float2; };
template <> struct FeatVec<float, 4> { using type = float4; };
template <> struct FeatVec<float, 8> { struct alignas(16) type { float4 a, b; }; };
template <> struct FeatVec<__half, 1> { using type = __half; };
template <> struct FeatVec<__half, 2> { using type = __half2; };
template <> struct FeatVec<__half, 4> { struct alignas(8) type { __half2 a, b; }; };
template <> struct FeatVec<__half, 8> { struct alignas(16) type { __half2 a, b, c, d; }; };

template <typename T, uint32_t C>
__device__ __forceinline__ void load_feat(const T* __restrict__ p, T (&out)[C]) {
    using V = typename FeatVec<T, C>::type;
    static_assert(sizeof(V) == sizeof(T) * C, "vector size");
    const V v = *reinterpret_cast<const V*>(p);
    __builtin_memcpy(out, &v, sizeof(V));
}
template <typename T, uint32_t C>
__device__ __forceinline__ void store_feat(T* __restrict__ p, const T (&in)[C]) {
    using V = typename FeatVec<T, C>::type;
    V v;
    __builtin_memcpy(&v, in, sizeof(V));
    *reinterpret_cast<V*>(p) = v;
}

template <typename T> struct Acc;
template <> struct Acc<float> {
    static __device__ __forceinline__ float zero() { return 0.0f; }
    // results += w * g  (fused, as nvcc contracts it)
    static __device__ __forceinline__ float fma(float w, float g, float acc) { return __builtin_fmaf(w, g, acc); }
    static __device__ __forceinline__ float sub(float a, float b) { return a - b; }
    static __device__ __forceinline__ float to_f(float a) { return a; }
    static __device__ __forceinline__ float from_f(float a) { return a; }
    static __device__ __forceinline__ float mul(float a, float b) { return a * b; }
    static __device__ __forceinline__ float add(float a, float b) { return a + b; }
};
template <> struct Acc<__half> {
    static __device__ __forceinline__ __half zero() { return __float2half(0.0f); }
    // at::Half += float : the float product is rounded to half, then a half add (gridencoder.cu:184)
    static __device__ __forceinline__ __half fma(float w, __half g, __half acc) {
        return __hadd(acc, __float2half(w * __half2float(g)));
    }
    static __device__ __forceinline__ __half sub(__half a, __half b) { return __hsub(a, b); }
    static __device__ __forceinline__ float to_f(__half a) { return __half2float(a); }
    static __device__ __forceinline__ __half from_f(float a) { return __float2half(a); }
    static __device__ __forceinline__ __half mul(__half a, __half b) { return __hmul(a, b); }
    static __device__ __forceinline__ __half add(__half a, __half b) { return __hadd(a, b); }
};

template <uint32_t D>
__device__ __forceinline__ bool load_point(const float* __restrict__ inputs, uint32_t b, float (&x)[D]) {
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) oob = true;
    }
    return oob;
}

template <uint32_t D>
__device__ __forceinline__ void locate(const float (&x)[D], float scale, bool align_corners, uint32_t interp,
                                       float (&pos)[D], float (&pos_deriv)[D], uint32_t (&pos_grid)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = __builtin_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            pos_deriv[d] = 6 * pos[d] * (1.0f - pos[d]);
            pos[d] = pos[d] * pos[d] * __builtin_fmaf(-2.0f, pos[d], 3.0f);
        }
    }
}

constexpr uint32_t kXcds = 8;
constexpr uint32_t kFwdBlock = 256;

// ------------------------------------------------------------------ forward
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kFwdBlock) k_grid_forward(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                            const int32_t* __restrict__ offsets, T* __restrict__ outputs,
                                                            uint32_t B, uint32_t L, LevelScales scales,
                                                            T* __restrict__ dy_dx, uint32_t gridtype, bool align_corners,
                                                            uint32_t interp) {
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t b = (blockIdx.x / kXcds) * kFwdBlock + threadIdx.x;
    if (b >= B || xcd >= L) return;
    float x[D];
    const bool oob = load_point<D>(inputs, b, x);

    for (uint32_t level = xcd; level < L; level += kXcds) {
        T* out = outputs + ((size_t)level * B + b) * C;
        T* jac = dy_dx ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;
        if (oob) {
            T z[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) z[c] = Acc<T>::zero();
            store_feat<T, C>(out, z);
            if (jac) {
#pragma unroll
                for (uint32_t d = 0; d < D; d++) store_feat<T, C>(jac + d * C, z);
            }
            continue;
        }
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        const T* table = grid + (size_t)off * C;
        const float scale = scales.v[level];
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        float pos[D], pos_deriv[D];
        uint32_t pos_grid[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) pos_deriv[d] = (d == 0) ? 1.0f : 0.0f;  // `= {1.0f}`, gridencoder.cu:143
        locate<D>(x, scale, align_corners, interp, pos, pos_deriv, pos_grid);

        // issue all 2^D gathers first, then reduce in corner order
        T feat[1u << D][C];
        float wts[1u << D];
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1;
            uint32_t pgl[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
            }
            wts[idx] = w;
            const uint32_t row = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
            load_feat<T, C>(table + (size_t)row * C, feat[idx]);
        }
        T res[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) res[c] = Acc<T>::zero();
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) res[c] = Acc<T>::fma(wts[idx], feat[idx][c], res[c]);
        }
        store_feat<T, C>(out, res);

        if (jac) {  // gridencoder.cu:198-241
#pragma unroll
            for (uint32_t gd = 0; gd < D; gd++) {
                T g[C];
#pragma unroll
                for (uint32_t c = 0; c < C; c++) g[c] = Acc<T>::zero();
#pragma unroll
                for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                    float w = scale;
                    uint32_t pgl[D];
#pragma unroll
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                        if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                        else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                    }
                    pgl[gd] = pos_grid[gd];
                    const uint32_t rl = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
                    pgl[gd] = pos_grid[gd] + 1;
                    const uint32_t rr = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
                    T fl[C], fr[C];
                    load_feat<T, C>(table + (size_t)rl * C, fl);
                    load_feat<T, C>(table + (size_t)rr * C, fr);
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) {
                        const float diff = Acc<T>::to_f(Acc<T>::sub(fr[c], fl[c]));
                        if constexpr (sizeof(T) == 2) g[c] = Acc<T>::add(g[c], Acc<T>::from_f(w * diff * pos_deriv[gd]));
                        else g[c] = Acc<T>::from_f(__builtin_fmaf(w * diff, pos_deriv[gd], Acc<T>::to_f(g[c])));
                    }
                }
                store_feat<T, C>(jac + gd * C, g);
            }
        }
    }
}

// test hook: rows of all corners
template <uint32_t D>
__global__ void k_grid_corner_rows(const float* __restrict__ inputs, const int32_t* __restrict__ offsets,
                                   uint32_t* __restrict__ corner_idx, uint32_t B, uint32_t L, LevelScales scales,
                                   uint32_t gridtype, bool align_corners) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    float x[D];
    const bool oob = load_point<D>(inputs, b, x);
    uint32_t* o = corner_idx + ((size_t)b * L + level) * (1u << D);
    if (oob) {
        for (uint32_t i = 0; i < (1u << D); i++) o[i] = 0xffffffffu;
        return;
    }
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = scales.v[level];
    const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
    float pos[D], pd[D];
    uint32_t pos_grid[D];
    locate<D>(x, scale, align_corners, 0, pos, pd, pos_grid);
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        uint32_t pgl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) pgl[d] = pos_grid[d] + ((idx >> d) & 1u);
        o[idx] = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
    }
}

// ------------------------------------------------------------------ backward
__device__ __forceinline__ void atomic_add_feat(float* p, float v) { unsafeAtomicAdd(p, v); }

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kFwdBlock) k_grid_backward(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                             const int32_t* __restrict__ offsets, T* __restrict__ grad_grid,
                                                             uint32_t B, uint32_t L, LevelScales scales, uint32_t gridtype,
                                                             bool align_corners, uint32_t interp) {
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t b = (blockIdx.x / kXcds) * kFwdBlock + threadIdx.x;
    if (b >= B || xcd >= L) return;
    float x[D];
    if (load_point<D>(inputs, b, x)) return;  // grad is zero-initialised by the caller

    for (uint32_t level = xcd; level < L; level += kXcds) {
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        T* table = grad_grid + (size_t)off * C;
        const float scale = scales.v[level];
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        float pos[D], pd[D];
        uint32_t pos_grid[D];
        locate<D>(x, scale, align_corners, interp, pos, pd, pos_grid);
        T g[C];
        load_feat<T, C>(grad + ((size_t)level * B + b) * C, g);
        // samples past a ray's early termination (and padding rows) carry exactly-zero gradients: adding zero is a
        // no-op, and atomics are the bottleneck (~21 G/s chip-wide), so skip them
        bool nonzero = false;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) nonzero |= (Acc<T>::to_f(g[c]) != 0.0f);
        if (!nonzero) continue;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1;
            uint32_t pgl[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
            }
            const uint32_t row = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pgl);
            T* dst = table + (size_t)row * C;
            if constexpr (sizeof(T) == 2) {
                // packed half2 atomics (global_atomic_pk_add_f16), gridencoder.cu:322-328
#pragma unroll
                for (uint32_t c = 0; c < C; c += 2) {
                    const __half2 v = __halves2half2(__float2half(w * __half2float(g[c])),
                                                     __float2half(w * __half2float(g[c + 1])));
                    unsafeAtomicAdd(reinterpret_cast<__half2*>(dst + c), v);
                }
            } else {
#pragma unroll
                for (uint32_t c = 0; c < C; c++) atomic_add_feat(reinterpret_cast<float*>(dst) + c, w * g[c]);
            }
        }
    }
}


// ---- LDS fixed-point backward -----------------------------------------------------------------------
// v * 2^k as a 64-bit integer: v = m * 2^(ex-24) with a 24-bit integer mantissa m, so the product is a shift
// (round-half-up when bits fall off the bottom).  Callers guarantee |v| * 2^k < 2^62.
__device__ __forceinline__ long long to_fixed64(float v, int k) {
    int ex;
    const float f = frexpf(v, &ex);
    const long long m = (long long)(int)ldexpf(f, 24);
    const int sh = k + ex - 24;
    if (sh >= 0) return m << sh;
    if (sh > -26) return (m + (1ll << (-sh - 1))) >> (-sh);
    return 0;
}
constexpr uint32_t kLdsBytes = 160 * 1024;
constexpr uint32_t kBwdThreads = 1024;
constexpr uint32_t kQueueLen = 192;                                   // per-wave hit queue entries (<64 pending + <=128 appended)
constexpr uint32_t kQueueBytes = (kBwdThreads / 64) * kQueueLen * 4;  // 8 KiB at the top of the LDS allocation
constexpr uint32_t kAccBytes = kLdsBytes - kQueueBytes;
// Every workgroup scans all points, so its cost is (scan) + (hits); hits per workgroup = B * 2^D / slices(level).
// Coarse levels have few rows: cutting them by LDS capacity alone would leave ONE workgroup with every hit of the
// level (measured: the level-0 workgroup set the kernel time).  Each level is therefore cut into at least
// kMinSlices slices — the slice count LDS capacity forces on a 2^19-row hashed level.
__host__ __device__ constexpr uint32_t bwd_min_slices(uint32_t C) { return div_up<uint32_t>(1u << 19, kAccBytes / (8 * C)); }
__host__ __device__ inline uint32_t bwd_slice_rows(uint32_t rows, uint32_t C) {
    const uint32_t cap = kAccBytes / (8 * C);
    uint32_t r = div_up<uint32_t>(div_up<uint32_t>(rows, bwd_min_slices(C)), 8u) * 8u;
    return r < cap ? (r ? r : 8u) : cap;
}

// Points are re-packed to one 16-byte record each before the sweep: measured on MI355X a wave-level vector load
// costs ~20 cycles of address-unit time per instruction regardless of width, and a 12-byte-stride [B,3] read is
// three of them per point (0.34 ns/point/CU) against 0.15 for one aligned float4 — and the sweep reads every point
// once per (level, slice) workgroup.  Out-of-range points are encoded as NaN.x so the sweep needs no range test.
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_pack_points(const float* __restrict__ inputs, const T* __restrict__ grad, uint32_t B,
                                                     uint32_t L, float4* __restrict__ packed, uint32_t* __restrict__ count) {
    // Record = (x, y, z, original point index as bits).  Points that cannot contribute are DROPPED here, once,
    // instead of being re-scanned by every (level, slice) workgroup: out-of-range points, and points whose
    // gradient is exactly zero on every level (samples behind a ray's early termination, padding rows).
    // Integer accumulation is order-independent, so compaction does not affect the (bit-reproducible) result.
    static_assert(D <= 3, "packed record holds up to 3 coordinates + the point index");
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    float x[3] = {0, 0, 0};
    bool keep = b < B;
    if (keep) {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) { x[d] = inputs[(size_t)b * D + d]; keep &= (x[d] >= 0 && x[d] <= 1); }
    }
    if (keep) {
        bool nz = false;
        for (uint32_t l = 0; l < L && !nz; l++) {
            T g[C];
            load_feat<T, C>(grad + ((size_t)l * B + b) * C, g);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) nz |= (Acc<T>::to_f(g[c]) != 0.0f);
        }
        keep = nz;
    }
    const unsigned long long m = __ballot(keep);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, 0, 64);
    if (keep) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        packed[base + rank] = make_float4(x[0], x[1], x[2], __uint_as_float(b));
    }
}

// per-level max |grad| (bit pattern of a non-negative float is monotone as uint32)
template <typename T>
__global__ void __launch_bounds__(256) k_grad_absmax(const T* __restrict__ grad, uint32_t per_level, uint32_t* __restrict__ out) {
    // 16-byte loads, block-level reduction, ONE atomic per workgroup (all workgroups of a level hit the same word)
    constexpr uint32_t V = 16 / sizeof(T);
    const T* g = grad + (size_t)blockIdx.y * per_level;
    float m = 0.0f;
    auto upd = [&](float v) { v = fabsf(v); m = (v > m || v != v) ? v : m; };  // NaN propagates
    const uint32_t nvec = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? per_level / V : 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
        const uint4 raw = reinterpret_cast<const uint4*>(g)[i];
        T v[V];
        __builtin_memcpy(v, &raw, 16);
#pragma unroll
        for (uint32_t k = 0; k < V; k++) upd(Acc<T>::to_f(v[k]));
    }
    for (uint32_t i = nvec * V + blockIdx.x * 256 + threadIdx.x; i < per_level; i += gridDim.x * 256) upd(Acc<T>::to_f(g[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m, d, 64); m = (o > m || o != o) ? o : m; }
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; w++) { const float o = part[w]; m = (o > m || o != o) ? o : m; }
        atomicMax(out + blockIdx.y, __float_as_uint(m));
    }
}

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kBwdThreads) k_grid_backward_lds(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                                  const float4* __restrict__ packed,
                                                                  const int32_t* __restrict__ offsets, T* __restrict__ grad_grid,
                                                                  uint32_t B, uint32_t L, LevelScales scales,
                                                                  const uint32_t* __restrict__ absmax, uint32_t gridtype,
                                                                  bool align_corners, uint32_t interp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem_raw);
    // (level, slice) of this workgroup from the device offset table; the host launches an upper bound
    // ceil(total_rows / slice) + L workgroups, the surplus exits here
    uint32_t level = 0, first = 0, slice_rows = 0;
    for (;; level++) {
        if (level == L) return;
        const uint32_t rows = (uint32_t)(offsets[level + 1] - offsets[level]);
        slice_rows = bwd_slice_rows(rows, C);
        const uint32_t ns = div_up<uint32_t>(rows, slice_rows);
        if (blockIdx.x < first + ns) break;
        first += ns;
    }
    const uint32_t slice = blockIdx.x - first;
    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
    const uint32_t row0 = slice * slice_rows;
    const uint32_t nrows = min(slice_rows, hashmap_size - row0);

    const uint32_t n_points = (D <= 3) ? absmax[kMaxLevels] : B;  // compacted point count (written by k_pack_points)
    const float amax = __uint_as_float(absmax[level]);
    if (!(amax > 0.0f)) {
        if (amax != amax || amax == INFINITY) {  // non-finite gradient: poison the slice like a float sum would
            for (uint32_t i = threadIdx.x; i < nrows * C; i += kBwdThreads)
                grad_grid[((size_t)off + row0) * C + i] = Acc<T>::from_f(NAN);
        }
        return;  // all-zero gradient: nothing to add
    }
    if (amax == INFINITY) {
        for (uint32_t i = threadIdx.x; i < nrows * C; i += kBwdThreads) grad_grid[((size_t)off + row0) * C + i] = Acc<T>::from_f(NAN);
        return;
    }
    // |sum| <= B * 2^D * amax < 2^62 : pick the power-of-two scale accordingly
    int e;
    (void)frexpf(amax, &e);  // amax < 2^e
    const int kexp = 62 - e - (int)(32 - __clz(B)) - (int)D;
    const double inv_scale = ldexp(1.0, -kexp);

    for (uint32_t i = threadIdx.x; i < nrows * C; i += kBwdThreads) acc[i] = 0ull;
    __syncthreads();

    const float lscale = scales.v[level];
    const uint32_t resolution = (uint32_t)ceilf(lscale) + 1;
    const T* glevel = grad + (size_t)level * B * C;

    // Level-uniform index plan, hoisted out of the point loop (get_grid_index, gridencoder.cu:66-84):
    // which dimensions enter the dense index (the stride loop stops once stride > hashmap_size), their strides,
    // and whether the level is hashed.  `row % hashmap_size` becomes a mask when the size is a power of two and a
    // no-op test for dense levels (index < rows by construction).
    uint32_t stride[D];
    uint32_t st = 1;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (st <= hashmap_size) { stride[d] = st; st *= align_corners ? resolution : (resolution + 1); }
        else stride[d] = 0;  // dimension dropped from the dense index
    }
    const bool hashed = (gridtype == 0 && st > hashmap_size);
    const bool pow2 = (hashmap_size & (hashmap_size - 1)) == 0;
    const uint32_t mask = hashmap_size - 1;

    // Hits are sparse (~2^D/slices per point) and scattered over lanes: executing the accumulate body under
    // divergence costs a wave-iteration per hit-bearing lane (measured 3x the scan itself).  Instead every wave
    // appends its hits (point, corner) to a small LDS queue (wave prefix sum of per-lane hit counts) and drains the
    // queue 64 entries at a time with all lanes busy.
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* queue = reinterpret_cast<uint32_t*>(smem_raw + kLdsBytes - kQueueBytes) + wave * kQueueLen;

    // The scan loop is specialised on the (workgroup-uniform) index mode so its body carries no selects:
    //   MODE 0: hashed level, power-of-two rows (row = xor-hash & mask)      — the eleven 2^19-row levels
    //   MODE 1: dense level (row = sum of strides; `% rows` only if index >= rows)
    //   MODE 2: anything else (generic get_grid_index semantics)
    auto row_of = [&](auto mode, const uint32_t (&lo)[D], const uint32_t (&hi)[D], uint32_t idx) -> uint32_t {
        constexpr int MODE = decltype(mode)::value;
        uint32_t index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const uint32_t t = ((idx >> d) & 1u) ? hi[d] : lo[d];
            if (MODE == 0) index ^= t;
            else if (MODE == 1) index += t;
            else index = hashed ? (index ^ t) : (index + t);
        }
        if (MODE == 0) return index & mask;
        if (MODE == 1) return (index < hashmap_size) ? index : index % hashmap_size;
        return pow2 ? (index & mask) : ((index < hashmap_size) ? index : index % hashmap_size);
    };
    auto corner_terms = [&](auto mode, const uint32_t (&pos_grid)[D], uint32_t (&lo)[D], uint32_t (&hi)[D]) {
        constexpr int MODE = decltype(mode)::value;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const bool h = (MODE == 0) || (MODE == 2 && hashed);
            if (h) { lo[d] = pos_grid[d] * kPrimes[d]; hi[d] = lo[d] + kPrimes[d]; }
            else { lo[d] = pos_grid[d] * stride[d]; hi[d] = lo[d] + stride[d]; }
        }
    };

    auto run = [&](auto mode) {
        uint32_t qlen = 0;  // wave-uniform
        // Drain, software-pipelined by one stage: a drain ISSUES the loads of its 64 entries (packed point + gradient
        // row) and PROCESSES the entries whose loads the previous drain issued, so the ~1-2 us round trip overlaps
        // the scan in between instead of stalling the wave (PMC: 51 % of wave cycles were s_waitcnt stalls).
        bool pend = false;           // this lane holds a loaded, unprocessed entry
        uint32_t pend_idx = 0, pend_b = 0;
        float pend_x[D];
        T pend_g[C];
        auto process = [&]() {
            if (pend) {
                float pos[D], pd[D];
                uint32_t pos_grid[D], lo[D], hi[D];
                locate<D>(pend_x, lscale, align_corners, interp, pos, pd, pos_grid);
                corner_terms(mode, pos_grid, lo, hi);
                const uint32_t local = row_of(mode, lo, hi, pend_idx) - row0;
                float w = 1;
#pragma unroll
                for (uint32_t d = 0; d < D; d++) w *= ((pend_idx >> d) & 1u) ? pos[d] : 1 - pos[d];
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    // same product the reference forms (float w * grad), then exact scaling to 64-bit fixed point
                    float prod;
                    if constexpr (sizeof(T) == 2) prod = __half2float(__float2half(w * __half2float(pend_g[c])));
                    else prod = w * pend_g[c];
                    atomicAdd(&acc[local * C + c], (unsigned long long)to_fixed64(prod, kexp));
                }
                pend = false;
            }
        };
        auto drain = [&](uint32_t first, uint32_t count) {  // lanes [0,count) take entries first..first+count-1
            process();
            if (lane < count) {
                const uint32_t ent = queue[first + lane];
                uint32_t b = ent >> D;  // slot in the packed array (D <= 3) or point index
                pend_idx = ent & ((1u << D) - 1);
                if constexpr (D <= 3) {
                    const float4 p = packed[b];
                    const float v[3] = {p.x, p.y, p.z};
#pragma unroll
                    for (uint32_t d = 0; d < D; d++) pend_x[d] = v[d];
                    b = __float_as_uint(p.w);  // original point index, for the gradient row
                } else {
#pragma unroll
                    for (uint32_t d = 0; d < D; d++) pend_x[d] = inputs[(size_t)b * D + d];
                }
                pend_b = b;
                load_feat<T, C>(glevel + (size_t)b * C, pend_g);
                pend = true;
            }
        };
        // Scan kUnroll points per lane, then append ALL their hits with one wave prefix sum: the hit test is pure
        // VALU, and the VALU->SGPR->branch round trips of the queue bookkeeping (ballots, popcounts, drain test) are
        // paid once per kUnroll points instead of once per point.
        constexpr uint32_t kCorners = 1u << D;
        constexpr uint32_t kUnroll = kCorners <= 8 ? 4 : (kCorners <= 16 ? 2 : 1);
        static_assert(kUnroll * kCorners <= 32, "per-lane hit mask must fit 32 bits");
        auto scan = [&](uint32_t b0, const float (&xs)[kUnroll][D]) {
            uint32_t hits = 0;  // bit (u * 2^D + corner)
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; u++) {
                bool oob = false;
#pragma unroll
                for (uint32_t d = 0; d < D; d++) oob |= !(xs[u][d] >= 0 && xs[u][d] <= 1);  // also true for the padding marker
                float pos[D], pd[D];
                uint32_t pos_grid[D], lo[D], hi[D];
                locate<D>(xs[u], lscale, align_corners, interp, pos, pd, pos_grid);
                corner_terms(mode, pos_grid, lo, hi);
                uint32_t h = 0;
#pragma unroll
                for (uint32_t idx = 0; idx < kCorners; idx++)
                    h |= ((row_of(mode, lo, hi, idx) - row0) < nrows ? 1u : 0u) << idx;
                hits |= (oob ? 0u : h) << (u * kCorners);
            }
            if (__ballot(hits != 0) == 0) return;
            // exclusive prefix sum of the per-lane hit counts from ballots of the count bits (no LDS round trips)
            const uint32_t cnt = __popc(hits);
            uint32_t excl = 0, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) {
                const unsigned long long mk = __ballot((cnt >> k) & 1u);
                excl += __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u)) << k;
                total += (uint32_t)__popcll(mk) << k;
            }
            if (total <= kQueueLen - 64) {
                uint32_t slot = qlen + excl;
                while (hits) {
                    const uint32_t j = __builtin_ctz(hits);
                    hits &= hits - 1;
                    queue[slot++] = ((b0 + (j / kCorners) * kBwdThreads) << D) | (j % kCorners);
                }
                qlen += total;
                while (qlen >= 64) { qlen -= 64; drain(qlen, 64); }
            } else {  // pathological density (e.g. many identical points): one (point, corner) at a time
#pragma unroll 1
                for (uint32_t j = 0; j < kUnroll * kCorners; j++) {
                    const bool hit = (hits >> j) & 1u;
                    const unsigned long long m = __ballot(hit);
                    if (hit) queue[qlen + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                        ((b0 + (j / kCorners) * kBwdThreads) << D) | (j % kCorners);
                    qlen += (uint32_t)__popcll(m);
                    if (qlen >= 64) { qlen -= 64; drain(qlen, 64); }
                }
            }
        };
        // All co-resident workgroups sweep the point array in the SAME order: the 32 CUs of an XCD then touch the
        // same window at about the same time and share it through their L2.  (PMC, profiles/r01: with per-workgroup
        // start offsets FETCH_SIZE was 2.8 GB per launch — 70 % of the 930 x 4.2 MB point re-reads missed the 4 MiB
        // L2 and the sweep ran at the fabric's ~2.4 TB/s.)
        const uint32_t per_round = kBwdThreads * kUnroll;
        const uint32_t rounds = div_up<uint32_t>(n_points, per_round);
        auto fetch = [&](uint32_t r, float (&xs)[kUnroll][D]) {
            const uint32_t b0 = r * per_round + threadIdx.x;
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; u++) {
                const uint32_t b = b0 + u * kBwdThreads;
                if constexpr (D <= 3) {
                    const float4 p = (b < n_points) ? packed[b] : make_float4(-1.0f, 0, 0, 0);
                    const float v[3] = {p.x, p.y, p.z};
#pragma unroll
                    for (uint32_t d = 0; d < D; d++) xs[u][d] = v[d];
                } else {
#pragma unroll
                    for (uint32_t d = 0; d < D; d++) xs[u][d] = (b < n_points) ? inputs[(size_t)b * D + d] : -1.0f;
                }
            }
            return b0;
        };
        // software pipeline: the loads of round r+1 are in flight while round r is scanned (PMC: 48 % of the wave
        // cycles were s_waitcnt stalls with the loads issued at the top of the same round)
        float cur[kUnroll][D], nxt[kUnroll][D];
        uint32_t b_cur = fetch(0, cur), b_nxt = 0;
        for (uint32_t r = 0; r < rounds; r++) {
            if (r + 1 < rounds) b_nxt = fetch(r + 1, nxt);
            scan(b_cur, cur);
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; u++)
#pragma unroll
                for (uint32_t d = 0; d < D; d++) cur[u][d] = nxt[u][d];
            b_cur = b_nxt;
        }
        drain(0, qlen);
        process();
    };
    bool all_dims = true;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) all_dims &= (stride[d] != 0);
    if (hashed && pow2) run(std::integral_constant<int, 0>{});
    else if (!hashed && all_dims) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    __syncthreads();
    T* dst = grad_grid + ((size_t)off + row0) * C;
    for (uint32_t i = threadIdx.x; i < nrows * C; i += kBwdThreads) {
        const long long q = (long long)acc[i];
        if (q != 0) {
            const float add = (float)((double)q * inv_scale);
            dst[i] = Acc<T>::from_f(Acc<T>::to_f(dst[i]) + add);
        }
    }
}

// gridencoder.cu:340-366
template <typename T, uint32_t D, uint32_t C>
__global__ void k_grid_input_backward(const T* __restrict__ grad, const T* __restrict__ dy_dx, T* __restrict__ grad_inputs,
                                      uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* j = dy_dx + (size_t)b * L * D * C;
    T r = Acc<T>::zero();
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const T gv = grad[((size_t)l * B + b) * C + c];
            const T jv = j[(size_t)l * D * C + d * C + c];
            if constexpr (sizeof(T) == 2) r = Acc<T>::add(r, Acc<T>::mul(gv, jv));
            else r = Acc<T>::from_f(__builtin_fmaf(Acc<T>::to_f(gv), Acc<T>::to_f(jv), Acc<T>::to_f(r)));
        }
    }
    grad_inputs[t] = r;
}

// ------------------------------------------------------------------ TV gradient (fp32)
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kFwdBlock) k_grad_tv(const float* __restrict__ inputs, const float* __restrict__ grid,
                                                       float* __restrict__ grad, const int32_t* __restrict__ offsets,
                                                       float weight, uint32_t B, uint32_t L, LevelScales scales,
                                                       uint32_t gridtype, bool align_corners) {
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t b = (blockIdx.x / kXcds) * kFwdBlock + threadIdx.x;
    if (b >= B || xcd >= L) return;
    float x[D];
    if (load_point<D>(inputs, b, x)) return;
    for (uint32_t level = xcd; level < L; level += kXcds) {
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        const float* table = grid + (size_t)off * C;
        float* gtable = grad + (size_t)off * C;
        const float scale = scales.v[level];
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        uint32_t pos_grid[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++)
            pos_grid[d] = (uint32_t)floorf(__builtin_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f));
        float results[C], idelta[C], center[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) { results[c] = 0; idelta[c] = 0; }
        const uint32_t row = grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid);
        load_feat<float, C>(table + (size_t)row * C, center);
        const float w = weight / (float)(2 * D);
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const uint32_t cur = pos_grid[d];
            if (cur < resolution) {
                pos_grid[d] = cur + 1;
                float nb[C];
                load_feat<float, C>(table + (size_t)grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, nb);
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    const float gv = center[c] - nb[c];
                    results[c] += gv;
                    idelta[c] = __builtin_fmaf(gv, gv, idelta[c]);
                }
            }
            if (cur > 0) {
                pos_grid[d] = cur - 1;
                float nb[C];
                load_feat<float, C>(table + (size_t)grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C, nb);
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    const float gv = center[c] - nb[c];
                    results[c] += gv;
                    idelta[c] = __builtin_fmaf(gv, gv, idelta[c]);
                }
            }
            pos_grid[d] = cur;
        }
#pragma unroll
        for (uint32_t c = 0; c < C; c++)
            atomic_add_feat(gtable + (size_t)row * C + c, w * results[c] * (1.0f / sqrtf(idelta[c] + 1e-9f)));
    }
}

void host_scales(uint32_t L, float S, uint32_t H, LevelScales& out) {
    for (uint32_t l = 0; l < kMaxLevels; l++) out.v[l] = 0.0f;
    for (uint32_t l = 0; l < L; l++) out.v[l] = fmaf(exp2f((float)l * S), (float)H, -1.0f);
}

inline uint32_t xcd_grid(uint32_t B) { return kXcds * div_up<uint32_t>(B, kFwdBlock); }

template <typename T, uint32_t D>
int launch_forward(const float* inputs, const T* emb, const int32_t* offsets, T* outputs, uint32_t B, uint32_t C,
                   uint32_t L, const LevelScales& sc, T* dy_dx, uint32_t gridtype, bool ac, uint32_t interp,
                   hipStream_t st) {
    const dim3 grid(xcd_grid(B)), block(kFwdBlock);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_forward<T, D, 1>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        case 2: hipLaunchKernelGGL((k_grid_forward<T, D, 2>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        case 4: hipLaunchKernelGGL((k_grid_forward<T, D, 4>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        case 8: hipLaunchKernelGGL((k_grid_forward<T, D, 8>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return S3D_ERR_UNSUPPORTED;
    }
    return check_launch("grid_encode_forward");
}

constexpr uint32_t kLdsBackwardMinPoints = 8192;  // below this the direct-atomic kernel wins (fixed sweep cost)

template <typename T, uint32_t D, uint32_t C>
int launch_backward_c(const T* grad, const float* inputs, const int32_t* offsets, uint32_t table_rows, T* grad_emb,
                      uint32_t B, uint32_t L, const LevelScales& sc, const T* dy_dx, T* grad_inputs, uint32_t gridtype, bool ac,
                      uint32_t interp, uint32_t* ws, size_t ws_bytes, int force_path, hipStream_t st) {
    const bool ws_ok = ws && ws_bytes >= 256 + (size_t)B * 16;
    const bool use_lds = (force_path == 2 && ws_ok && table_rows) || (force_path == 0 && B >= kLdsBackwardMinPoints && table_rows && ws_ok);
    if (use_lds) {
        // upper bound on the number of (level, slice) workgroups: capacity slices + per-level minimum + remainders
        const uint32_t nb = div_up<uint32_t>(table_rows, kAccBytes / (8 * C)) + L * (bwd_min_slices(C) + 1);
        S3D_HIP(hipMemsetAsync(ws, 0, sizeof(uint32_t) * (kMaxLevels + 1), st));
        float4* packed = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(ws) + 256);
        if constexpr (D <= 3)
            hipLaunchKernelGGL((k_pack_points<T, D, C>), dim3(div_up<uint32_t>(B, 256)), dim3(256), 0, st, inputs, grad, B, L,
                               packed, ws + kMaxLevels);
        const uint32_t per_level = B * C;
        uint32_t gx = div_up<uint32_t>(per_level, 256 * 32);
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL((k_grad_absmax<T>), dim3(gx, L), dim3(256), 0, st, grad, per_level, ws);
        static bool attr_set = false;
        if (!attr_set) {
            S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_backward_lds<T, D, C>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
            attr_set = true;
        }
        hipLaunchKernelGGL((k_grid_backward_lds<T, D, C>), dim3(nb), dim3(kBwdThreads), kLdsBytes, st, grad, inputs,
                           (const float4*)packed, offsets, grad_emb, B, L, sc, (const uint32_t*)ws, gridtype, ac, interp);
    } else {
        hipLaunchKernelGGL((k_grid_backward<T, D, C>), dim3(xcd_grid(B)), dim3(kFwdBlock), 0, st, grad, inputs, offsets,
                           grad_emb, B, L, sc, gridtype, ac, interp);
    }
    if (dy_dx && grad_inputs)
        hipLaunchKernelGGL((k_grid_input_backward<T, D, C>), dim3(div_up<uint32_t>(B * D, 256)), dim3(256), 0, st, grad,
                           dy_dx, grad_inputs, B, L);
    return check_launch("grid_encode_backward");
}

template <typename T, uint32_t D>
int launch_backward(const T* grad, const float* inputs, const int32_t* offsets, uint32_t table_rows, T* grad_emb,
                    uint32_t B, uint32_t C, uint32_t L, const LevelScales& sc, const T* dy_dx, T* grad_inputs, uint32_t gridtype,
                    bool ac, uint32_t interp, uint32_t* ws, size_t ws_bytes, int force_path, hipStream_t st) {
    switch (C) {
        case 1:
            if constexpr (sizeof(T) == 2) {
                set_error("GridEncoding: fp16 tables need an even C (the reference forces fp32 when C is odd, grid.py:42)");
                return S3D_ERR_UNSUPPORTED;
            } else return launch_backward_c<T, D, 1>(grad, inputs, offsets, table_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, st);
        case 2: return launch_backward_c<T, D, 2>(grad, inputs, offsets, table_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, st);
        case 4: return launch_backward_c<T, D, 4>(grad, inputs, offsets, table_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, st);
        case 8: return launch_backward_c<T, D, 8>(grad, inputs, offsets, table_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, st);
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return S3D_ERR_UNSUPPORTED;
    }
}

template <uint32_t D>
int launch_tv(const float* inputs, const float* emb, float* grad, const int32_t* offsets, float weight, uint32_t B,
              uint32_t C, uint32_t L, const LevelScales& sc, uint32_t gridtype, bool ac, hipStream_t st) {
    const dim3 grid(xcd_grid(B)), block(kFwdBlock);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grad_tv<D, 1>), grid, block, 0, st, inputs, emb, grad, offsets, weight, B, L, sc, gridtype, ac); break;
        case 2: hipLaunchKernelGGL((k_grad_tv<D, 2>), grid, block, 0, st, inputs, emb, grad, offsets, weight, B, L, sc, gridtype, ac); break;
        case 4: hipLaunchKernelGGL((k_grad_tv<D, 4>), grid, block, 0, st, inputs, emb, grad, offsets, weight, B, L, sc, gridtype, ac); break;
        case 8: hipLaunchKernelGGL((k_grad_tv<D, 8>), grid, block, 0, st, inputs, emb, grad, offsets, weight, B, L, sc, gridtype, ac); break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return S3D_ERR_UNSUPPORTED;
    }
    return check_launch("grad_total_variation");
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT void s3d_grid_level_scales(uint32_t L, float S, uint32_t H, float* scales_out) {
    LevelScales sc;
    host_scales(L > kMaxLevels ? kMaxLevels : L, S, H, sc);
    for (uint32_t l = 0; l < L && l < kMaxLevels; l++) scales_out[l] = sc.v[l];
}

#define S3D_DISPATCH_D(D, CALL2, CALL3, CALL4, CALL5)                                     \
    switch (D) {                                                                          \
        case 2: return CALL2;                                                             \
        case 3: return CALL3;                                                             \
        case 4: return CALL4;                                                             \
        case 5: return CALL5;                                                             \
        default: set_error("GridEncoding: D must be 2, 3, 4, or 5."); return S3D_ERR_UNSUPPORTED; \
    }

S3D_EXPORT int s3d_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                                       void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                       void* dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                       s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: null pointer");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward: L must be in [1, %u]", kMaxLevels);
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "grid_encode_forward: dtype must be f32 or f16");
    S3D_REQUIRE((uint64_t)B * L * C < (1ull << 32), "grid_encode_forward: B*L*C overflows 32 bits");
    LevelScales sc;
    host_scales(L, S, H, sc);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    if (dtype == S3D_F32) {
        const float* e = (const float*)embeddings; float* o = (float*)outputs; float* j = (float*)dy_dx;
        S3D_DISPATCH_D(D, (launch_forward<float, 2>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)),
                       (launch_forward<float, 3>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)),
                       (launch_forward<float, 4>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)),
                       (launch_forward<float, 5>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)))
    } else {
        const __half* e = (const __half*)embeddings; __half* o = (__half*)outputs; __half* j = (__half*)dy_dx;
        S3D_DISPATCH_D(D, (launch_forward<__half, 2>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)),
                       (launch_forward<__half, 3>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)),
                       (launch_forward<__half, 4>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)),
                       (launch_forward<__half, 5>(inputs, e, offsets, o, B, C, L, sc, j, gridtype, ac, interp, st)))
    }
}

template <uint32_t D>
static int launch_corner_rows(const float* inputs, const int32_t* offsets, uint32_t* corner_idx, uint32_t B, uint32_t L,
                              const LevelScales& sc, uint32_t gridtype, bool ac, hipStream_t st) {
    hipLaunchKernelGGL((k_grid_corner_rows<D>), dim3(div_up<uint32_t>(B, 256), L), dim3(256), 0, st, inputs, offsets,
                       corner_idx, B, L, sc, gridtype, ac);
    return check_launch("grid_corner_indices");
}

S3D_EXPORT int s3d_grid_corner_indices(const float* inputs, const int32_t* offsets, uint32_t* corner_idx, uint32_t B,
                                       uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                       int align_corners, s3d_stream_t stream) {
    (void)C;
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && offsets && corner_idx, "grid_corner_indices: null pointer");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_corner_indices: L must be in [1, %u]", kMaxLevels);
    LevelScales sc;
    host_scales(L, S, H, sc);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    S3D_DISPATCH_D(D, (launch_corner_rows<2>(inputs, offsets, corner_idx, B, L, sc, gridtype, ac, st)),
                   (launch_corner_rows<3>(inputs, offsets, corner_idx, B, L, sc, gridtype, ac, st)),
                   (launch_corner_rows<4>(inputs, offsets, corner_idx, B, L, sc, gridtype, ac, st)),
                   (launch_corner_rows<5>(inputs, offsets, corner_idx, B, L, sc, gridtype, ac, st)))
}

S3D_EXPORT size_t s3d_grid_encode_backward_workspace_size(uint32_t B) { return 256 + (size_t)B * 16; }

// process-wide override for experiments/tests: 0 = auto, 1 = direct atomics, 2 = LDS fixed-point sweep
static int g_backward_path = 0;
S3D_EXPORT void s3d_grid_backward_set_path(int path) { g_backward_path = path; }

S3D_EXPORT int s3d_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                        const int32_t* offsets, void* grad_embeddings, uint32_t table_rows,
                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                        uint32_t interp, int dtype, void* workspace, size_t workspace_bytes,
                                        s3d_stream_t stream) {
    (void)embeddings;
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: null pointer");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward: L must be in [1, %u]", kMaxLevels);
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "grid_encode_backward: dtype must be f32 or f16");
    S3D_REQUIRE(g_backward_path != 2 || (workspace && table_rows && workspace_bytes >= s3d_grid_encode_backward_workspace_size(B)),
                "grid_encode_backward: LDS path needs table_rows and a workspace of s3d_grid_encode_backward_workspace_size(B) bytes");
    LevelScales sc;
    host_scales(L, S, H, sc);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    uint32_t* ws = (uint32_t*)workspace;
    const int fp = g_backward_path;
    if (dtype == S3D_F32) {
        const float* g = (const float*)grad; float* ge = (float*)grad_embeddings;
        const float* j = (const float*)dy_dx; float* gi = (float*)grad_inputs;
        S3D_DISPATCH_D(D, (launch_backward<float, 2>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)),
                       (launch_backward<float, 3>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)),
                       (launch_backward<float, 4>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)),
                       (launch_backward<float, 5>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)))
    } else {
        const __half* g = (const __half*)grad; __half* ge = (__half*)grad_embeddings;
        const __half* j = (const __half*)dy_dx; __half* gi = (__half*)grad_inputs;
        S3D_DISPATCH_D(D, (launch_backward<__half, 2>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)),
                       (launch_backward<__half, 3>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)),
                       (launch_backward<__half, 4>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)),
                       (launch_backward<__half, 5>(g, inputs, offsets, table_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, st)))
    }
}

S3D_EXPORT int s3d_grad_total_variation(const float* inputs, const float* embeddings, float* grad,
                                        const int32_t* offsets, float weight, uint32_t B, uint32_t D, uint32_t C,
                                        uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                        s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && embeddings && grad && offsets, "grad_total_variation: null pointer");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grad_total_variation: L must be in [1, %u]", kMaxLevels);
    LevelScales sc;
    host_scales(L, S, H, sc);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    S3D_DISPATCH_D(D, (launch_tv<2>(inputs, embeddings, grad, offsets, weight, B, C, L, sc, gridtype, ac, st)),
                   (launch_tv<3>(inputs, embeddings, grad, offsets, weight, B, C, L, sc, gridtype, ac, st)),
                   (launch_tv<4>(inputs, embeddings, grad, offsets, weight, B, C, L, sc, gridtype, ac, st)),
                   (launch_tv<5>(inputs, embeddings, grad, offsets, weight, B, C, L, sc, gridtype, ac, st)))
}
