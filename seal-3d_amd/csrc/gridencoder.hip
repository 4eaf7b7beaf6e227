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
#include "s3d_common.hpp"
#include <math.h>

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

template <typename T, uint32_t D, uint32_t C>
int launch_backward_c(const T* grad, const float* inputs, const int32_t* offsets, T* grad_emb, uint32_t B, uint32_t L,
                      const LevelScales& sc, const T* dy_dx, T* grad_inputs, uint32_t gridtype, bool ac, uint32_t interp,
                      hipStream_t st) {
    hipLaunchKernelGGL((k_grid_backward<T, D, C>), dim3(xcd_grid(B)), dim3(kFwdBlock), 0, st, grad, inputs, offsets,
                       grad_emb, B, L, sc, gridtype, ac, interp);
    if (dy_dx && grad_inputs)
        hipLaunchKernelGGL((k_grid_input_backward<T, D, C>), dim3(div_up<uint32_t>(B * D, 256)), dim3(256), 0, st, grad,
                           dy_dx, grad_inputs, B, L);
    return check_launch("grid_encode_backward");
}

template <typename T, uint32_t D>
int launch_backward(const T* grad, const float* inputs, const int32_t* offsets, T* grad_emb, uint32_t B, uint32_t C,
                    uint32_t L, const LevelScales& sc, const T* dy_dx, T* grad_inputs, uint32_t gridtype, bool ac,
                    uint32_t interp, hipStream_t st) {
    switch (C) {
        case 1:
            if constexpr (sizeof(T) == 2) {
                set_error("GridEncoding: fp16 tables need an even C (the reference forces fp32 when C is odd, grid.py:42)");
                return S3D_ERR_UNSUPPORTED;
            } else return launch_backward_c<T, D, 1>(grad, inputs, offsets, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 2: return launch_backward_c<T, D, 2>(grad, inputs, offsets, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 4: return launch_backward_c<T, D, 4>(grad, inputs, offsets, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 8: return launch_backward_c<T, D, 8>(grad, inputs, offsets, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, st);
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

S3D_EXPORT int s3d_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                        const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                        uint32_t L, float S, uint32_t H, const void* dy_dx, void* grad_inputs,
                                        uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                        s3d_stream_t stream) {
    (void)embeddings;
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: null pointer");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward: L must be in [1, %u]", kMaxLevels);
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "grid_encode_backward: dtype must be f32 or f16");
    LevelScales sc;
    host_scales(L, S, H, sc);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    if (dtype == S3D_F32) {
        const float* g = (const float*)grad; float* ge = (float*)grad_embeddings;
        const float* j = (const float*)dy_dx; float* gi = (float*)grad_inputs;
        S3D_DISPATCH_D(D, (launch_backward<float, 2>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)),
                       (launch_backward<float, 3>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)),
                       (launch_backward<float, 4>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)),
                       (launch_backward<float, 5>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)))
    } else {
        const __half* g = (const __half*)grad; __half* ge = (__half*)grad_embeddings;
        const __half* j = (const __half*)dy_dx; __half* gi = (__half*)grad_inputs;
        S3D_DISPATCH_D(D, (launch_backward<__half, 2>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)),
                       (launch_backward<__half, 3>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)),
                       (launch_backward<__half, 4>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)),
                       (launch_backward<__half, 5>(g, inputs, offsets, ge, B, C, L, sc, j, gi, gridtype, ac, interp, st)))
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
