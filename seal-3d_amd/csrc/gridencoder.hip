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
//  * Backward: direct atomics for small batches; for training-size batches the contributions are partitioned by
//    table slice and summed in LDS as 64-bit fixed point (see "binned backward" below): deterministic, no global atomics.
#include "s3d_common.hpp"
#include "s3d_adam.hpp"
#include <math.h>
#include <type_traits>
#include <algorithm>

namespace s3d {
namespace {

constexpr uint32_t kMaxLevels = 32;
// per-level scales + the optional input normalisation of GridEncoder.forward (grid.py:146: x01 = (x + bound) / (2 bound),
// evaluated as torch's GPU kernels do: one add, one multiply by the fp32 reciprocal); bound = 0: inputs are already in [0,1]
struct LevelScales {
    float v[kMaxLevels];
    float bound, inv_2bound;
    const int32_t* n_valid;   // see valid_rows()
    const float* live;        // forward only: rows with live[b * live_stride] == 0 are written as zeros, table untouched
    uint32_t live_stride;
};

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
    // `index % hashmap_size` without the ~25-instruction division by a run-time value on the paths that never need it:
    // hashed levels have power-of-two sizes (a mask), dense rows lie below the size (nothing to do); what is left (tiled
    // grids whose strides overflow the table) divides
    const uint32_t mask = hashmap_size - 1;
    if ((hashmap_size & mask) == 0) return index & mask;
    if (__builtin_expect(index >= hashmap_size, 0)) index %= hashmap_size;
    return index;
}

// Level-uniform index plan (get_grid_index, gridencoder.cu:66-84): which dimensions enter the dense index (the
// stride loop stops once stride > hashmap_size), their strides, whether the level is hashed; `% hashmap_size` is a
// mask for power-of-two sizes and a no-op for dense rows below the size.  Same rows as grid_row, fewer divisions.
template <uint32_t D>
struct LevelIndex {
    uint32_t mul[D];  // per-dimension multiplier: prime (hashed) or stride (dense; 0 = dimension dropped)
    uint32_t size, mask;
    bool hashed, pow2, need_mod;
    __device__ __forceinline__ void init(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution) {
        uint32_t st = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if (st <= hashmap_size) { mul[d] = st; st *= align_corners ? resolution : (resolution + 1); }
            else mul[d] = 0;
        }
        hashed = (gridtype == 0 && st > hashmap_size);
        if (hashed) {
#pragma unroll
            for (uint32_t d = 0; d < D; d++) mul[d] = kPrimes[d];
        }
        size = hashmap_size;
        mask = hashmap_size - 1;
        pow2 = (hashmap_size & mask) == 0;
        // can a dense index reach the table size at all?  (cell coordinates are <= resolution + 1)
        unsigned long long top = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) top += (unsigned long long)(resolution + 1) * mul[d];
        need_mod = hashed || top >= hashmap_size;
    }
    __device__ __forceinline__ uint32_t row(const uint32_t (&lo)[D], uint32_t idx) const {  // lo[d] = pos_grid[d] * mul[d]
        uint32_t index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const uint32_t t = ((idx >> d) & 1u) ? lo[d] + mul[d] : lo[d];
            index = hashed ? (index ^ t) : (index + t);
        }
        if (pow2) return index & mask;
        if (!need_mod) return index;  // (wave-uniform: dense levels skip the division by a run-time value altogether)
        return index < size ? index : index % size;
    }
};

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
__device__ __forceinline__ bool load_point(const float* __restrict__ inputs, uint32_t b, const LevelScales& sc, float (&x)[D]) {
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (sc.bound != 0.0f) x[d] = (x[d] + sc.bound) * sc.inv_2bound;
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
    if (b >= valid_rows(B, scales.n_valid) || xcd >= L) return;
    float x[D];
    bool oob = load_point<D>(inputs, b, scales, x);
    // unused sample slots of the inference loop (march_rays leaves them zero-filled, deltas == 0; composite_rays never
    // reads their sigma / rgb): same treatment as out-of-range points — zeros, no gathers
    if (scales.live && scales.live[(size_t)b * scales.live_stride] == 0.0f) oob = true;

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

// ---- forward, lane pairs -------------------------------------------------------------------------------------
// The two x-corners of a (y,z,..) corner pair are neighbouring rows: always on dense levels, and on hashed levels too —
// the hash is `x ^ (y * p1) ^ (z * p2)`, so for fixed (y, z) the 32 rows of a 128-byte line (fp16, C = 2) are 32
// consecutive x cells.  With one lane per point the two x-corners are fetched by two DIFFERENT gather instructions, each
// of which presents 64 distinct lines to the L1 / L2 (the measured ceiling of this kernel is the request rate, not
// bytes).  Here lanes 2k and 2k+1 share their two points P, Q: in the first half of a level's gathers both lanes address
// point P (lane 2k its x0 corners, lane 2k+1 its x1 corners — the same lines, inside one instruction, which the
// address unit merges), in the second half point Q; the halves a lane fetched for its partner travel by DPP
// (quad_perm [1,0,3,2], no LDS), and each lane then reduces ITS point in the reference's corner order: same values bit
// for bit, half the distinct lines per gather instruction.
__device__ __forceinline__ uint32_t dpp_swap1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // lane ^ 1
}
__device__ __forceinline__ float dpp_swap1(float v) { return __uint_as_float(dpp_swap1(__float_as_uint(v))); }

// Which levels workgroup (xcd, chunk) serves: bit l of mask[xcd][chunk % kFwdResidues].  Every (level, chunk) pair belongs to
// exactly one XCD (balance_forward_plan() below): a level's table stays in the L2 of the few XCDs that serve it.
constexpr uint32_t kFwdResidues = 16;
constexpr uint32_t kFwdMaxChunks = 2048;  // chunk slots per XCD of a launch (a multiple of kFwdResidues)
struct FwdPlan {
    uint32_t mask[kXcds][kFwdResidues];
};
// [0]: the home mapping (level l on XCD l % 8), [1]: the balanced plan for coherent points.  Every WAVE picks one for its 64
// points from the points themselves (the same decision in all eight workgroups that hold these points), see the kernel.
struct FwdPlans {
    FwdPlan p[2];
};

template <typename T, uint32_t D, uint32_t C, bool STRIDED = false>
__global__ void __launch_bounds__(kFwdBlock) k_grid_forward_pair(const float* __restrict__ inputs, const T* __restrict__ grid_a,
                                                                 const int32_t* __restrict__ offsets, T* __restrict__ outputs_a,
                                                                 uint32_t B, uint32_t L, LevelScales scales, uint32_t gridtype,
                                                                 bool align_corners, uint32_t interp, FwdPlans plans,
                                                                 const T* __restrict__ grid_b, T* __restrict__ outputs_b) {
    static_assert((sizeof(T) * C) % 4 == 0, "feature vectors travel between lanes as 32-bit words");
    // blockIdx.y = 1: the SECOND table of a two-encoder call on the same points (s3d_grid_encode_forward_pair: the density and
    // the colour encoder of the network Seal-3D trains share geometry and inputs — one launch, one ramp and one tail for both)
    const T* __restrict__ grid = blockIdx.y ? grid_b : grid_a;
    T* __restrict__ outputs = blockIdx.y ? outputs_b : outputs_a;
    constexpr uint32_t NW = sizeof(T) * C / 4;   // words per feature vector
    constexpr uint32_t J = 1u << (D - 1);        // corner pairs per point
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t Bv = valid_rows(B, scales.n_valid);
    // STRIDED (batches of more than kFwdMaxChunks chunks outside the inference loop): the launch holds kFwdMaxChunks chunk slots
    // per XCD and a workgroup walks the chunks slot, slot + slots, ... while they hold valid rows.  (A padded batch of
    // N x max_steps rows with 2e5 of them filled — the teacher's proxy render of the Seal step — dispatched 250,000 workgroups
    // that left at once: 30 of that launch's 113 us.)  One workgroup per chunk otherwise: chunks dispatched in order keep
    // neighbouring rays in flight together (a fully valid 2^21-point batch loses 6 % when it is walked in strides).
    uint32_t chunk = blockIdx.x / kXcds;
    do {
    const uint32_t b = chunk * kFwdBlock + threadIdx.x;
    // a wave leaves as a whole (its lanes exchange data below): Bv is a multiple of the block size or the last block is ragged
    if ((b & ~63u) >= Bv) return;  // (rows ascend with the chunk: nothing behind this one either)
    const bool valid = b < Bv;
    const uint32_t side = threadIdx.x & 1u;  // which x-corner this lane fetches, for both points of the pair
    float x[D];
    bool oob = true;
    if (valid) {
        oob = load_point<D>(inputs, b, scales, x);
        if (scales.live && scales.live[(size_t)b * scales.live_stride] == 0.0f) oob = true;
    } else {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) x[d] = 0.0f;
    }
    const bool oob_partner = dpp_swap1(oob ? 1u : 0u) != 0u;
    const bool oobP = side ? oob_partner : oob, oobQ = side ? oob : oob_partner;
    // Which plan serves these 64 points?  The balanced plan pays off on coherent points (samples along rays, Morton- or
    // lattice-ordered cells: coarse levels are cheap there, so their XCDs take slices of the busy ones) and costs ~50 % on
    // scattered points (every level costs the same: the home mapping IS balanced).  Coherent = a quarter or more of the neighbours in the wave lie
    // within 0.02 (L1, unit cube) of each other — a pure function of the wave's points, so the eight workgroups that hold
    // them (one per XCD) agree, and each (level, 64 points) is still served exactly once.
    float dl1 = 0.0f;
#pragma unroll
    for (uint32_t d = 0; d < D; d++)
        dl1 += fabsf(x[d] - __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x[d]), 0x138, 0xF, 0xF, true)));  // wave_shr:1
    const unsigned long long near_mask = __ballot(valid && (threadIdx.x & 63u) != 0u && dl1 < 0.02f);
    // (a `live` mask comes with the inference loop's slot rows — n_step consecutive slots per ray, unfilled ones parked at the
    //  origin: ray-ordered by construction, though the parked rows break the neighbour test)
    const uint32_t coherent = (scales.live != nullptr || __popcll(near_mask) >= 16) ? 1u : 0u;  // (wave-uniform)
    uint32_t todo = plans.p[coherent].mask[xcd][chunk % kFwdResidues];  // (wave-uniform) this wave's levels
    if (todo == 0u) {
        if constexpr (STRIDED) continue;
        else return;
    }

    struct LevelState {
        float pos[D];
        uint32_t g1[J][NW], g2[J][NW];  // this lane's x-side of point P / of point Q
    };
    // phase 1 of a level: cell + weights of the own point, rows of the lane's x-side for both points, gathers issued
    auto issue = [&](uint32_t level, LevelState& st) {
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        const T* table = grid + (size_t)off * C;
        const float scale = scales.v[level];
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        float pos_deriv[D];
        uint32_t pg[D], pg_partner[D];
        locate<D>(x, scale, align_corners, interp, st.pos, pos_deriv, pg);
#pragma unroll
        for (uint32_t d = 0; d < D; d++) pg_partner[d] = dpp_swap1(pg[d]);
        // level-uniform index plan: per-dimension multipliers, hashed or dense, mask or (rarely) a division — decided once
        // per level on the scalar unit, so that the J gathers of a half are straight-line code issued back to back
        LevelIndex<D> li;
        li.init(gridtype, align_corners, hashmap_size, resolution);
        auto gather = [&](auto hashed_c, const uint32_t (&tg)[D], bool skip, uint32_t (&dst)[J][NW]) {
            constexpr bool kHashed = decltype(hashed_c)::value;
            uint32_t lo[D], hi[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) { lo[d] = tg[d] * li.mul[d]; hi[d] = lo[d] + li.mul[d]; }
            const uint32_t x0 = side ? hi[0] : lo[0];
            uint32_t rows[J];
#pragma unroll
            for (uint32_t j = 0; j < J; j++) {
                uint32_t index = x0;
#pragma unroll
                for (uint32_t d = 1; d < D; d++) {
                    const uint32_t t = ((j >> (d - 1)) & 1u) ? hi[d] : lo[d];
                    index = kHashed ? (index ^ t) : (index + t);
                }
                rows[j] = index;
            }
            if (li.pow2) {
#pragma unroll
                for (uint32_t j = 0; j < J; j++) rows[j] &= li.mask;
            } else if (li.need_mod) {
#pragma unroll
                for (uint32_t j = 0; j < J; j++) rows[j] = rows[j] < li.size ? rows[j] : rows[j] % li.size;
            }
#pragma unroll
            for (uint32_t j = 0; j < J; j++)
#pragma unroll
                for (uint32_t k = 0; k < NW; k++) dst[j][k] = 0u;
            if (!skip) {
#pragma unroll
                for (uint32_t j = 0; j < J; j++) {
                    T f[C];
                    load_feat<T, C>(table + (size_t)rows[j] * C, f);
                    __builtin_memcpy(dst[j], f, sizeof(T) * C);
                }
            }
        };
        // first half: both lanes address point P (the even lane's point), second half: point Q
        uint32_t tgP[D], tgQ[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) { tgP[d] = side ? pg_partner[d] : pg[d]; tgQ[d] = side ? pg[d] : pg_partner[d]; }
        // samples marched along a ray stay in one cell of a coarse level for many consecutive samples: when P and Q share
        // their cell, Q's corners ARE P's — the second half's gathers are masked off and the values copied
        bool same = !oobP && !oobQ;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) same = same && (tgP[d] == tgQ[d]);
        if (li.hashed) {
            gather(std::true_type{}, tgP, oobP, st.g1);
            gather(std::true_type{}, tgQ, oobQ || same, st.g2);
        } else {
            gather(std::false_type{}, tgP, oobP, st.g1);
            gather(std::false_type{}, tgQ, oobQ || same, st.g2);
        }
        if (same) {
#pragma unroll
            for (uint32_t j = 0; j < J; j++)
#pragma unroll
                for (uint32_t k = 0; k < NW; k++) st.g2[j][k] = st.g1[j][k];
        }
    };
    // phase 2: the halves fetched for the partner change lanes, the own point is reduced in the reference's corner order
    auto finish = [&](uint32_t level, const LevelState& st) {
        // lane 2k keeps g1 (P, x0) and needs lane 2k+1's g1 (P, x1); lane 2k+1 keeps g2 (Q, x1) and needs lane 2k's g2 (Q, x0)
        T feat[1u << D][C];
#pragma unroll
        for (uint32_t j = 0; j < J; j++) {
            uint32_t own[NW], got[NW];
#pragma unroll
            for (uint32_t k = 0; k < NW; k++) {
                own[k] = side ? st.g2[j][k] : st.g1[j][k];
                got[k] = dpp_swap1(side ? st.g1[j][k] : st.g2[j][k]);
            }
            uint32_t f0[NW], f1[NW];
#pragma unroll
            for (uint32_t k = 0; k < NW; k++) { f0[k] = side ? got[k] : own[k]; f1[k] = side ? own[k] : got[k]; }
            __builtin_memcpy(feat[(j << 1)], f0, sizeof(T) * C);
            __builtin_memcpy(feat[(j << 1) | 1u], f1, sizeof(T) * C);
        }
        if (!valid) return;
        T* out = outputs + ((size_t)level * B + b) * C;
        T res[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) res[c] = Acc<T>::zero();
        if (!oob) {
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
#pragma unroll
                for (uint32_t d = 0; d < D; d++) w *= ((idx >> d) & 1u) ? st.pos[d] : 1 - st.pos[d];
#pragma unroll
                for (uint32_t c = 0; c < C; c++) res[c] = Acc<T>::fma(w, feat[idx][c], res[c]);
            }
        }
        store_feat<T, C>(out, res);
    };
    // two levels in flight at a time: the second level's gathers are issued before the first one's are awaited
    (void)L;
    while (todo) {
        LevelState s0, s1;
        const uint32_t l0 = (uint32_t)__builtin_ctz(todo);
        todo &= todo - 1u;
        const bool two = todo != 0u;  // (uniform)
        const uint32_t l1 = two ? (uint32_t)__builtin_ctz(todo) : 0u;
        if (two) todo &= todo - 1u;
        issue(l0, s0);
        if (two) issue(l1, s1);
        finish(l0, s0);
        if (two) finish(l1, s1);
    }
    } while (STRIDED && ((chunk += gridDim.x / kXcds), true));
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
    const bool oob = load_point<D>(inputs, b, scales, x);
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
    if (b >= valid_rows(B, scales.n_valid) || xcd >= L) return;
    float x[D];
    if (load_point<D>(inputs, b, scales, x)) return;  // grad is zero-initialised by the caller

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


// ---- binned backward: partition the contributions by table slice, accumulate each slice in LDS --------------
// v * 2^k rounded to the nearest integer (ties to even), |v| * 2^k < 2^62, in fp32 and 32-bit integer ops only:
// t = |v| * 2^(k-32) is exact, floor(t) is the high word, the (exact) remainder scaled by 2^32 and rounded is the low word.
// [Two other forms were tried.  (a) frexp mantissa moved with variable 64-bit shifts inside divergent branches: wrong
//  values for a few lanes per million when inlined next to ds_add_u64 on gfx950 / ROCm 7.2, although the same source is
//  exact in a plain store kernel.  (b) __double2ll_rn(ldexp((double)v, k)): exact, but ~12 half-rate fp64 instructions
//  per value were ~half of the accumulate kernel's time.  tools/ubench/fix64_lds.hip reproduces (a) and verifies this
//  form bit for bit against llrint(ldexp((double)v, k)) for several k.]
__device__ __forceinline__ long long to_fixed64(float v, int k) {
    const float t = ldexpf(fabsf(v), k - 32);
    const float hf = floorf(t);
    const uint32_t hi = (uint32_t)(int)hf;
    const uint32_t lo = (uint32_t)rintf(ldexpf(t - hf, 32));
    const long long q = (long long)(((unsigned long long)hi << 32) | lo);
    return v < 0 ? -q : q;
}

// Measured on MI355X (tools/ubench): global fp atomics retire at a flat ~21 G/s chip-wide whatever the locality, LDS
// float atomics at ~0.2 T/s, LDS *integer* atomics at ~2.3 T/s.  So the table gradient is accumulated in LDS as
// 64-bit fixed point, one workgroup per (level, table slice), and the (point, level, corner) contributions are first
// PARTITIONED by slice so that each workgroup streams exactly its own records:
//   k_bin_count    per-level |grad| max (fixes the fixed-point scale) + records per (level, slice)
//   k_bin_scatter  one workgroup = one chunk of points at one level, one lane = four consecutive points: forms the 2^D
//                  records {row-in-slice, sum of w * grad rounded to T} of every run of same-cell points (bin_quad),
//                  ranks them per slice with LDS counters, reserves a run in every slice's bucket (one returning
//                  atomic per slice) and stores the records at bucket start + reserved offset + rank
//   k_bin_accumulate  one workgroup = one (level, slice): streams its bucket, ds_add_u64 into the slice, adds the
//                  slice into the table with coalesced stores.
// Each contribution is computed once and costs at most one HBM write + one read of an 8-byte record (fp16, C = 2); integer
// adds commute, so the result is bit-reproducible run to run (the reference's float atomics are not) and carries
// ~40 bits below the largest |grad| of the level.  Small batches keep the direct-atomic kernel.
// [An earlier version skipped the partition: every (level, slice) workgroup scanned ALL points and kept the hits.
//  That is ~55x redundant index arithmetic; the partitioned form measured 2.3-3.5x faster at every size.]
//
// Slices are INTERLEAVED in groups of kBinGroup rows (slice = (row / kBinGroup) % S): a dense level's rows follow the
// scene's geometry, and contiguous slices would leave the slabs that cover the object with most of the records.
#ifndef S3D_BIN_CHUNK  // points per k_bin_scatter workgroup: every workgroup ends with one returning atomic per slice on the
#define S3D_BIN_CHUNK 1024  // same cursor words (measured: 1024 beats 2048 and 4096 — the reservations are not what bounds the kernel)
#endif
#ifndef S3D_BIN_ACC_KB  // tuning knobs of k_bin_accumulate (tools/build_variants.sh builds variants)
#define S3D_BIN_ACC_KB 128
#endif
#ifndef S3D_BIN_ACC_THREADS
#define S3D_BIN_ACC_THREADS 1024
#endif
#ifndef S3D_BIN_ACC_UNROLL
#define S3D_BIN_ACC_UNROLL 4
#endif
// log2 of the rows per interleave group.  Round 6: 5 -> 8.  With the table's Adam inside the accumulate a slice's optimizer state is
// read and written in pieces of (group rows x 8 B): 256-byte pieces at 32 rows, 2 KiB at 256 — scatter + accumulate-with-Adam
// 151 - 158 -> 144 - 145 us at 2.8e5 points (64 rows: 148, 128: 148, 512: 148 - 153 with the plain pair at 108, 1,024: 187;
// profiles/r11_grid_backward.md).  The plain pair does not care (97 - 98 us at 32 .. 256).
#ifndef S3D_BIN_GROUP_LOG
#define S3D_BIN_GROUP_LOG 8
#endif
constexpr uint32_t kBinGroupLog = S3D_BIN_GROUP_LOG;
constexpr uint32_t kBinGroup = 1u << kBinGroupLog;
#ifndef S3D_BIN_ACC_PER_CU  // persistent accumulate workgroups per CU (2 needs S3D_BIN_ACC_KB <= 64)
#define S3D_BIN_ACC_PER_CU 1
#endif
constexpr uint32_t kBinAccPerCu = S3D_BIN_ACC_PER_CU;
constexpr uint32_t kBinAccBytes = S3D_BIN_ACC_KB * 1024;
constexpr uint32_t kBinAccThreads = S3D_BIN_ACC_THREADS;
constexpr uint32_t kBinAccUnroll = S3D_BIN_ACC_UNROLL;
constexpr uint32_t kBinMinSlices = 64;
constexpr uint32_t kBinMaxSlices = 512;

// slices of a level: a power of two (slice / local-row arithmetic is shifts and masks), >= kBinMinSlices for balance
__host__ __device__ inline uint32_t bin_slices(uint32_t rows, uint32_t C) {
    const uint32_t cap = kBinAccBytes / (8 * C);  // rows per slice; a multiple of kBinGroup
    const uint32_t need = div_up<uint32_t>(div_up<uint32_t>(rows, kBinGroup) * kBinGroup, cap);
    uint32_t s = kBinMinSlices;
    while (s < need) s <<= 1;
    return s;
}
__host__ __device__ inline uint32_t bin_local_rows(uint32_t rows, uint32_t S) {
    return div_up<uint32_t>(div_up<uint32_t>(rows, kBinGroup), S) * kBinGroup;
}
constexpr uint32_t kBinQuad = 4;  // consecutive points handled (and merged) by one lane of k_bin_count / k_bin_scatter
template <typename T, uint32_t D, uint32_t C>
__host__ __device__ constexpr uint32_t bin_chunk_points() { return S3D_BIN_CHUNK; }  // points per scatter workgroup (lanes x 4)

template <typename T, uint32_t C>
__device__ __forceinline__ float absmax_feat(const T (&g)[C], bool& nonzero) {
    float m = 0.0f;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) {
        const float v = fabsf(Acc<T>::to_f(g[c]));
        nonzero |= (v != 0.0f);
        m = (v > m || v != v) ? v : m;  // NaN propagates
    }
    return m;
}

// Contributions of kBinQuad CONSECUTIVE points at one level, merged where neighbours share a cell.  Samples marched along
// a ray stay in one cell of a level for (cell width / step) consecutive samples — 37 on level 0 of the Lego config, still
// ~2 on level 9 — and then hit the same 2^D rows: their weighted gradients w * grad are summed here in fp32 and leave as ONE
// record per corner, rounded to T once (a single-point run is exactly the product the reference adds).  ~45 % fewer records on ray-ordered
// samples: less HBM traffic in both directions and far fewer same-row LDS conflicts in k_bin_accumulate.  The walk is
// deterministic (fixed quads, fixed order), so the result stays reproducible bit for bit.
//   emit(q, idx, slice, local_row, float sum[C]) is called once per corner of each run; q is the compile-time slot (the
//   quad position at which the run was closed), so callers can keep per-record state in registers.
template <typename T, uint32_t D, uint32_t C, typename Emit>
__device__ __forceinline__ void bin_quad(const float (&x)[kBinQuad][D], const T (&g)[kBinQuad][C], const bool (&in)[kBinQuad],
                                         float lscale, bool align_corners, uint32_t interp, const LevelIndex<D>& li, uint32_t S,
                                         uint32_t sshift, Emit&& emit) {
    constexpr uint32_t K = 1u << D;
    bool open = false;
    uint32_t cell[D], lo[D];
    float acc[K][C];
    auto flush = [&](auto slot) {
#pragma unroll
        for (uint32_t idx = 0; idx < K; idx++) {
            const uint32_t row = li.row(lo, idx);
            const uint32_t grp = row / kBinGroup;
            emit(slot, idx, grp & (S - 1), (grp >> sshift) * kBinGroup + row % kBinGroup, acc[idx]);
        }
    };
    auto step = [&](auto qc) {
        constexpr uint32_t q = decltype(qc)::value;
        bool nz = false;
        (void)absmax_feat<T, C>(g[q], nz);
        if (!(in[q] && nz)) return;
        float pos[D], pd[D];
        uint32_t pos_grid[D];
        locate<D>(x[q], lscale, align_corners, interp, pos, pd, pos_grid);
        bool same = open;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) same = same && (pos_grid[d] == cell[d]);
        if (!same) {
            if constexpr (q > 0) {
                if (open) flush(std::integral_constant<uint32_t, (q > 0 ? q - 1 : 0)>{});
            }
            open = true;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) { cell[d] = pos_grid[d]; lo[d] = pos_grid[d] * li.mul[d]; }
#pragma unroll
            for (uint32_t idx = 0; idx < K; idx++)
#pragma unroll
                for (uint32_t c = 0; c < C; c++) acc[idx][c] = 0.0f;
        }
        float gf[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) gf[c] = Acc<T>::to_f(g[q][c]);
#pragma unroll
        for (uint32_t idx = 0; idx < K; idx++) {
            float w = 1;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) w *= ((idx >> d) & 1u) ? pos[d] : 1 - pos[d];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) acc[idx][c] += w * gf[c];  // fp32 sum of the run's products; rounded to T once, at flush
        }
    };
    step(std::integral_constant<uint32_t, 0>{});
    step(std::integral_constant<uint32_t, 1>{});
    step(std::integral_constant<uint32_t, 2>{});
    step(std::integral_constant<uint32_t, 3>{});
    if (open) flush(std::integral_constant<uint32_t, kBinQuad - 1>{});
}

__global__ void __launch_bounds__(1024) k_zero_words(uint32_t* __restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) p[i] = 0;
}

// hdr[level] = max |grad| over in-range points (bit pattern of a non-negative float is monotone as uint32; NaN
// patterns sort above +inf); tot[level * smax + slice] = records of the slice.
// Few, fat workgroups: every workgroup ends each level with one global atomic per slice on the SAME smax words, and
// same-address device atomics serialise (measured: one workgroup per 256 points spent 190 us in them at B = 2^18).
constexpr uint32_t kBinCountThreads = 1024;
constexpr uint32_t kBinCountChunks = 64;  // workgroups per level
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kBinCountThreads) k_bin_count(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                                const int32_t* __restrict__ offsets, uint32_t B,
                                                                uint32_t points_per_block, LevelScales scales,
                                                                uint32_t* __restrict__ hdr, uint32_t* __restrict__ tot,
                                                                uint32_t smax, uint32_t gridtype, bool align_corners,
                                                                uint32_t interp) {
    __shared__ uint32_t cnt[kBinMaxSlices];
    __shared__ float wmax[kBinCountThreads / 64];
    const uint32_t b_begin = blockIdx.x * points_per_block;
    const uint32_t b_end = min(valid_rows(B, scales.n_valid), b_begin + points_per_block);
    const uint32_t level = blockIdx.y;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t S = bin_slices(hashmap_size, C);
    const float lscale = scales.v[level];
    LevelIndex<D> li;
    li.init(gridtype, align_corners, hashmap_size, (uint32_t)ceilf(lscale) + 1);
    for (uint32_t s = threadIdx.x; s < S; s += kBinCountThreads) cnt[s] = 0;
    __syncthreads();
    float m = 0.0f;
    const uint32_t sshift = 31 - __clz(S);
    for (uint32_t b0 = b_begin + threadIdx.x * kBinQuad; b0 < b_end; b0 += kBinQuad * kBinCountThreads) {
        float x[kBinQuad][D];
        T g[kBinQuad][C];
        bool in[kBinQuad];
#pragma unroll
        for (uint32_t q = 0; q < kBinQuad; q++) {
            const uint32_t b = b0 + q;
            in[q] = b < b_end;
#pragma unroll
            for (uint32_t c = 0; c < C; c++) g[q][c] = Acc<T>::zero();
            if (in[q]) {
                in[q] = !load_point<D>(inputs, b, scales, x[q]);
                load_feat<T, C>(grad + ((size_t)level * B + b) * C, g[q]);
                if (in[q]) {
                    bool nz = false;
                    const float a = absmax_feat<T, C>(g[q], nz);
                    m = (a > m || a != a) ? a : m;
                }
            }
        }
        bin_quad<T, D, C>(x, g, in, lscale, align_corners, interp, li, S, sshift,
                          [&](auto, uint32_t, uint32_t slice, uint32_t, const float (&)[C]) { atomicAdd(&cnt[slice], 1u); });
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m, d, 64); m = (o > m || o != o) ? o : m; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < kBinCountThreads / 64; w++) { const float o = wmax[w]; m = (o > m || o != o) ? o : m; }
        if (m != 0.0f) atomicMax(hdr + level, __float_as_uint(m));
    }
    for (uint32_t s = threadIdx.x; s < S; s += kBinCountThreads)
        if (cnt[s]) atomicAdd(&tot[level * smax + s], cnt[s]);
}

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(S3D_BIN_CHUNK / 4) k_bin_scatter(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                    const int32_t* __restrict__ offsets, uint32_t B, uint32_t level0,
                                                    LevelScales scales, const uint32_t* __restrict__ hdr,
                                                    const uint32_t* __restrict__ tot, uint32_t* __restrict__ cursor,
                                                    uint32_t smax, uint32_t* __restrict__ gkeys,
                                                    typename FeatVec<T, C>::type* __restrict__ gvals, uint32_t gridtype,
                                                    bool align_corners, uint32_t interp) {
    using V = typename FeatVec<T, C>::type;
    constexpr uint32_t K = 1u << D;
    constexpr uint32_t P = bin_chunk_points<T, D, C>();  // points per workgroup
    constexpr uint32_t NT = P / kBinQuad;                // lanes: each walks kBinQuad consecutive points
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem_raw);
    uint32_t* gbase = cnt + kBinMaxSlices;
    // 4-byte values (fp16 C=2, fp32 C=1) travel as ONE 8-byte {key, value} record: half the store / load instructions
    constexpr bool kPacked = sizeof(V) == 4;

    // all global loads of the prologue are issued before the first test (see k_bin_accumulate)
    const uint32_t level = level0 + blockIdx.y;
    const float amax = __uint_as_float(hdr[level]);
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t Bv = valid_rows(B, scales.n_valid);
    if (blockIdx.x * P >= Bv) return;  // (uniform) chunk entirely in the absent tail of a padded batch
    const uint32_t b0 = blockIdx.x * P + threadIdx.x * kBinQuad;
    float x[kBinQuad][D];
    T g[kBinQuad][C];
    bool in[kBinQuad];
#pragma unroll
    for (uint32_t q = 0; q < kBinQuad; q++) {
        in[q] = b0 + q < Bv;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) g[q][c] = Acc<T>::zero();
        if (in[q]) {
            in[q] = !load_point<D>(inputs, b0 + q, scales, x[q]);
            load_feat<T, C>(grad + ((size_t)level * B + b0 + q) * C, g[q]);
        }
    }
    constexpr uint32_t kPer = kBinMaxSlices / 64;
    uint32_t my_tot[kPer];  // first wave: the level's slice totals, lane t holds slices t, t + 64, t + 128, ...
    if (threadIdx.x < 64) {
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t sl = j * 64 + threadIdx.x;
            my_tot[j] = sl < smax ? tot[level * smax + sl] : 0u;
        }
    }
    if (!(amax > 0.0f) || amax == INFINITY) return;  // zero / non-finite levels carry no records (uniform exit)
    const uint32_t S = bin_slices(hashmap_size, C);
    const uint32_t sshift = 31 - __clz(S);
    const float lscale = scales.v[level];
    LevelIndex<D> li;
    li.init(gridtype, align_corners, hashmap_size, (uint32_t)ceilf(lscale) + 1);

    for (uint32_t s = threadIdx.x; s < S; s += NT) cnt[s] = 0;
    __syncthreads();

    // records of this lane: slot (q, idx) is filled when a run is closed at quad position q (at most one run per q)
    uint32_t key[kBinQuad][K], rank[kBinQuad][K];
    V val[kBinQuad][K];
    uint32_t closed = 0;  // bit q: slot row q holds a run
    bin_quad<T, D, C>(x, g, in, lscale, align_corners, interp, li, S, sshift,
                      [&](auto qc, uint32_t idx, uint32_t slice, uint32_t local, const float (&sum)[C]) {
                          constexpr uint32_t q = decltype(qc)::value;
                          closed |= 1u << q;
                          T pr[C];
#pragma unroll
                          for (uint32_t c = 0; c < C; c++) pr[c] = Acc<T>::from_f(sum[c]);
#pragma unroll
                          for (uint32_t i = 0; i < K; i++)  // (idx is a loop index of an unrolled loop: resolve it statically)
                              if (i == idx) {
                                  key[q][i] = (slice << 20) | local;
                                  rank[q][i] = atomicAdd(&cnt[slice], 1u);
                                  __builtin_memcpy(&val[q][i], pr, sizeof(V));
                              }
                      });
    __syncthreads();
    const uint32_t per = S / 64;  // S is a power of two >= 64
    // the first wave reserves the chunk's run in every slice's bucket: bucket start (exclusive scan of the level's slice
    // totals) + one returning atomic per non-empty slice
    if (threadIdx.x < 64) {
        uint32_t carry = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            if (j < per) {
                const uint32_t sl = j * 64 + threadIdx.x, c = cnt[sl];
                const uint32_t incl = wave_incl_scan(my_tot[j]);
                gbase[sl] = c ? carry + incl - my_tot[j] + atomicAdd(&cursor[level * smax + sl], c) : 0u;
                carry += __shfl(incl, 63, 64);
            }
        }
    }
    __syncthreads();  // gbase complete
    // Records go straight from registers to their place in the bucket (bucket start + the chunk's reserved run + rank).
    // [A 64 KiB LDS staging buffer that sorted the chunk by slice for contiguous stores was dropped with the quad merge:
    //  sized for the worst case (no merging) it left only 8 waves per CU and the kernel became latency-bound.  Each lane's
    //  8-byte stores are scattered, but a bucket's lines fill within one workgroup's run and merge in L2.]
    const size_t region = (size_t)blockIdx.y * K * B;  // this level's record region inside the pass
#pragma unroll
    for (uint32_t q = 0; q < kBinQuad; q++) {
        if (!((closed >> q) & 1u)) continue;
#pragma unroll
        for (uint32_t idx = 0; idx < K; idx++) {
            const uint32_t sl = key[q][idx] >> 20;
            const size_t at = region + gbase[sl] + rank[q][idx];
            if constexpr (kPacked) {
                uint32_t bits;
                __builtin_memcpy(&bits, &val[q][idx], 4);
                reinterpret_cast<uint2*>(gkeys)[at] = make_uint2(key[q][idx] & 0xfffffu, bits);
            } else {
                gkeys[at] = key[q][idx] & 0xfffffu;
                gvals[at] = val[q][idx];
            }
        }
    }
}

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kBinAccThreads) k_bin_accumulate(const uint32_t* __restrict__ gkeys,
                                                                  const typename FeatVec<T, C>::type* __restrict__ gvals,
                                                                  const int32_t* __restrict__ offsets, T* __restrict__ grad_grid,
                                                                  uint32_t B, uint32_t level0, const uint32_t* __restrict__ hdr,
                                                                  const uint32_t* __restrict__ tot, uint32_t smax) {
    using V = typename FeatVec<T, C>::type;
    constexpr uint32_t K = 1u << D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem_raw);  // [C][local_rows]: a wave's adds of one channel spread over all banks
    __shared__ uint32_t wsum[kBinAccThreads / 64];
    const uint32_t level = level0 + blockIdx.y, slice = blockIdx.x;  // slice < smax (grid), so every load below is in range
    // Every global value the prologue needs is loaded up front, before any of them is tested: a dependent round trip
    // costs ~2 us here (the words were just written by the previous kernel on another XCD), and this workgroup's
    // whole useful life is ~10 us.
    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t rows = (uint32_t)offsets[level + 1] - off;
    const float amax = __uint_as_float(hdr[level]);
    const uint32_t count = tot[level * smax + slice];
    uint32_t part = 0;  // bucket start = records of the slices before this one
    for (uint32_t s = threadIdx.x; s < slice; s += kBinAccThreads) part += tot[level * smax + s];
    const uint32_t S = bin_slices(rows, C);
    if (slice >= S) return;
    const uint32_t local_rows = bin_local_rows(rows, S);
    auto row_of_local = [&](uint32_t local) { return ((local / kBinGroup) * S + slice) * kBinGroup + local % kBinGroup; };
    T* table = grad_grid + (size_t)off * C;
    if (!(amax > 0.0f) || amax == INFINITY) {
        if (amax != amax || amax == INFINITY) {  // non-finite gradient: poison the level like a float sum would
            for (uint32_t i = threadIdx.x; i < local_rows * C; i += kBinAccThreads) {
                const uint32_t row = row_of_local(i / C);
                if (row < rows) table[(size_t)row * C + i % C] = Acc<T>::from_f(NAN);
            }
        }
        return;
    }
    if (count == 0) return;
    for (uint32_t i = threadIdx.x; i < local_rows * C; i += kBinAccThreads) acc[i] = 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    uint32_t start = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBinAccThreads / 64; w++) start += wsum[w];
    const uint32_t end = start + count;
    int e;
    (void)frexpf(amax, &e);  // amax < 2^e ; |sum| <= B * 2^D * amax < 2^62
    const int kexp = 62 - e - (int)(32 - __clz(B)) - (int)D;
    const double inv_scale = ldexp(1.0, -kexp);
    const size_t region = (size_t)blockIdx.y * K * B;
    const uint32_t* keys = gkeys + region;
    const V* vals = gvals + region;
    // stream the bucket: U independent record loads per lane in flight, then the adds
    constexpr uint32_t U = kBinAccUnroll;
    for (uint32_t i0 = start + threadIdx.x; i0 < end; i0 += U * kBinAccThreads) {
        uint32_t k[U];
        V v[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = i0 + u * kBinAccThreads;
            if (i < end) {
                if constexpr (sizeof(V) == 4) {
                    const uint2 r = reinterpret_cast<const uint2*>(gkeys)[region + i];
                    k[u] = r.x;
                    __builtin_memcpy(&v[u], &r.y, 4);
                } else { k[u] = keys[i]; v[u] = vals[i]; }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            if (i0 + u * kBinAccThreads < end) {
                T pr[C];
                __builtin_memcpy(pr, &v[u], sizeof(V));
#pragma unroll
                for (uint32_t c = 0; c < C; c++)
                    atomicAdd(&acc[c * local_rows + k[u]], (unsigned long long)to_fixed64(Acc<T>::to_f(pr[c]), kexp));
            }
        }
    }
    __syncthreads();
    // add the slice into the table: one vector load + store per touched row, W rows per lane in flight
    constexpr uint32_t W = C <= 2 ? 8 : (C == 4 ? 4 : 2);
    for (uint32_t r0 = threadIdx.x; r0 < local_rows; r0 += W * kBinAccThreads) {
        long long q[W][C];
        V old[W];
        bool nz[W];
#pragma unroll
        for (uint32_t w = 0; w < W; w++) {
            const uint32_t r = r0 + w * kBinAccThreads;
            nz[w] = false;
            if (r < local_rows) {
#pragma unroll
                for (uint32_t c = 0; c < C; c++) { q[w][c] = (long long)acc[c * local_rows + r]; nz[w] |= (q[w][c] != 0); }
            }
        }
#pragma unroll
        for (uint32_t w = 0; w < W; w++)
            if (nz[w]) old[w] = *reinterpret_cast<const V*>(table + (size_t)row_of_local(r0 + w * kBinAccThreads) * C);
#pragma unroll
        for (uint32_t w = 0; w < W; w++) {
            if (nz[w]) {
                T o[C];
                __builtin_memcpy(o, &old[w], sizeof(V));
#pragma unroll
                for (uint32_t c = 0; c < C; c++) o[c] = Acc<T>::from_f(Acc<T>::to_f(o[c]) + (float)((double)q[w][c] * inv_scale));
                store_feat<T, C>(table + (size_t)row_of_local(r0 + w * kBinAccThreads) * C, o);
            }
        }
    }
}

// ---- exact fixed-point sums (fp16, one 32-bit value word per record) --------------------------------------------
// The three-kernel path above (count -> scatter -> accumulate, a counting pass, 4 + elem * C byte records) stays the path
// of the WIDE records (fp32 C >= 2, fp16 C >= 4).  Where a record's value is one 32-bit word (fp16 C = 2 under -O, fp32
// C = 1) the kernels further down replace it.  What they share:
//  * fp16 gradients need no maximum: every binary16 value (subnormals included) is an integer multiple of 2^-24, so with
//    the FIXED scale 2^24 the LDS sum is EXACT (|v| * 2^24 < 2^40; 2^D * B <= 2^23 contributions keep it below 2^63).
//    fp32 gradients (and fp16 batches beyond that bound) keep the data-dependent scale: one streaming max pass
//    (k_bin_amax: no index arithmetic).
//  * buckets of FIXED capacity (2x the no-merge mean of a uniformly hit level; real levels merge on top), so no counting
//    pass; a scatter workgroup sorts the records of its chunk by slice in LDS, reserves its run in every slice's bucket and
//    writes each run with consecutive lanes on consecutive addresses.  What does not fit a bucket (samples clustered in a
//    few coarse cells) is SPILLED to the chunk's own region and listed per slice: exact and reproducible in every case.
// (The 8-byte-record pair k_bin_scatter4 / k_bin_accumulate4 of round 2 — `path = 3` — left the library in round 4; it is
//  in the history at ace1d33.)
constexpr uint32_t kFixedExp = 24;

// fixed-point sum -> float: the two 32-bit limbs of the MAGNITUDE converted separately and joined by one fma (|error| <= 1 ulp
// of the fp32 result, which is then rounded to the table's type), sign restored last — a handful of instructions instead of
// the int64 -> double -> float route.  (Limbs of the two's-complement value would not do: a small negative sum has a low
// limb just below 2^32, whose conversion to float rounds the whole value away.)
__device__ __forceinline__ float fixed_to_float(long long q, int kexp) {
    const unsigned long long a = q < 0 ? 0ull - (unsigned long long)q : (unsigned long long)q;
    const float v = ldexpf(__builtin_fmaf((float)(uint32_t)(a >> 32), 4294967296.0f, (float)(uint32_t)a), -kexp);
    return q < 0 ? -v : v;
}

// v * 2^24 as a 64-bit integer for a value that IS a multiple of 2^-24 below 2^16 in magnitude (every binary16): the scaled
// value is exact in fp32, and so are its two 16-bit-apart limbs — two truncating conversions, no rounding anywhere.
__device__ __forceinline__ long long fixed24_exact(float v) {
    const float s = v * 16777216.0f;                       // |s| < 2^40, exact
    const int a = (int)(s * 1.52587890625e-05f);            // trunc(s / 2^16), |a| < 2^24
    const int b = (int)__builtin_fmaf(-(float)a, 65536.0f, s);  // remainder, same sign, |b| < 2^16
    return ((long long)a << 16) + (long long)b;
}

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(1024) k_bin_amax(const T* __restrict__ grad, const float* __restrict__ inputs, uint32_t B,
                                                   LevelScales scales, uint32_t* __restrict__ hdr) {
    __shared__ float wmax[16];
    const uint32_t level = blockIdx.y;
    const uint32_t Bv = valid_rows(B, scales.n_valid);
    float m = 0.0f;
    for (uint32_t b = blockIdx.x * 1024 + threadIdx.x; b < Bv; b += gridDim.x * 1024) {
        float x[D];
        if (load_point<D>(inputs, b, scales, x)) continue;
        T g[C];
        load_feat<T, C>(grad + ((size_t)level * B + b) * C, g);
        bool nz = false;
        const float a = absmax_feat<T, C>(g, nz);
        m = (a > m || a != a) ? a : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m, d, 64); m = (o > m || o != o) ? o : m; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < 16; w++) { const float o = wmax[w]; m = (o > m || o != o) ? o : m; }
        if (m != 0.0f) atomicMax(hdr + level, __float_as_uint(m));
    }
}

// ---- binned backward, third generation -------------------------------------------------------------------------
// Measured on round 2's 8-byte-record kernels (profiles/r07_timed_region.md): the records were written once and read once — 548 B per
// point, as much as the whole algorithmic budget of the op — and the scatter spent more than half of its wave-cycles parked
// (one returning global atomic per slice in the middle of every workgroup's phase chain, every barrier draining the
// vector-memory queue).  This generation keeps the structure (partition by table slice, exact 64-bit fixed-point sums in
// LDS) and cuts the bytes and the bubbles:
//  * 6-byte records in two streams: a 16-bit row-in-slice and the 32-bit value word.  The accumulate reads four records
//    with one 8-byte and one 16-byte load.
//  * one lane = one point, and the run merge is WAVE-wide: consecutive samples of a ray share a cell of a coarse level for
//    tens of samples (37 on level 0 of the Lego configuration, ~2 on level 9); a head-flagged segmented scan sums the 2^D x C
//    products of a run across its lanes in fp32 (fixed tree, deterministic) and the run's LAST lane emits one record per
//    corner.  The scan stops as soon as no run is open (one ballot per wave on the fine levels, where nothing merges).
//  * XCD-private sub-buckets: a (level, slice) bucket is split in 8; a workgroup appends to the part of the XCD it runs on
//    (HW_REG_XCC_ID), so every 64-byte line of a bucket is written by ONE L2 and leaves it whole — runs of a few dozen
//    records from different XCDs no longer share lines.  Placement changes speed only: any sub-bucket choice gives the same
//    sums (integer adds commute).
//  * the bucket reservations (returning global atomics) are issued, the workgroup stages its records in LDS meanwhile
//    (barriers wait for LDS only), and the results are consumed just before the copy-out.
#ifndef S3D_BIN3_P
#define S3D_BIN3_P 512
#endif
#ifndef S3D_BIN3_NSUB   // XCD-private sub-buckets per (level, slice): 8x fewer same-word cursor atomics (r08: 152 -> 131 us), 1 = off
#define S3D_BIN3_NSUB 8
#endif
#ifndef S3D_BIN3_MIN_MERGES  // fewer continuing lanes than this in a wave: no merge there (the scan costs more than it saves)
#define S3D_BIN3_MIN_MERGES 6
#endif
#ifndef S3D_BIN3_CURSOR_STRIDE  // words between two cursors (16 = one cursor per 64-byte line)
#define S3D_BIN3_CURSOR_STRIDE 16
#endif
constexpr uint32_t kCurStride = S3D_BIN3_CURSOR_STRIDE;
#ifdef S3D_BIN3_PROF  // per-workgroup phase stamps (shader clock) of the two kernels: tools/debug only
__device__ unsigned long long s3d_prof_buf[2][16384][8];
#define S3D_STAMP(kern, wg, k) do { if (threadIdx.x == 0 && (wg) < 16384) s3d_prof_buf[kern][wg][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S3D_STAMP(kern, wg, k) do { } while (0)
#endif
static_assert(kBinGroup == (1u << kBinGroupLog) && kBinGroupLog >= 5 && kBinGroupLog <= 12, "the scatter's key arithmetic shifts by kBinGroupLog");
constexpr uint32_t kBin3Sub = S3D_BIN3_NSUB;
static_assert(kBin3Sub == 1 || kBin3Sub == 2 || kBin3Sub == 4 || kBin3Sub == 8, "sub-buckets follow the XCD id");

// workgroup barrier that waits for this wave's LDS traffic only (a __syncthreads() also drains the vector-memory queue)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

// ---- wave-wide segmented scan by DPP (no LDS traffic) ----
// f = 1 once the lane has absorbed its run's head.  In-row steps (row_shr 1, 2, 4, 8 inside the 16-lane rows), then the two
// carry steps of a wave scan (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): a lane takes the carry only
// while it is still open, and inherits the flag of the lane it took from.  Fixed tree: the fp32 sums are deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_take(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// One step = ONE instruction per value: v += dpp(v) * m with m = 1.0 on the lanes that are still open and 0.0 on the closed
// ones (v_fmac_f32 with a DPP source; lanes without a source lane read 0, rows outside row_mask are not written), and
// f |= dpp(f).  u * 1 + v is the fp32 sum bit for bit; a closed lane next to a non-finite product turns NaN (0 * inf),
// which poisons the level exactly as that product does on its own.  Inline asm: four values per statement behind one
// `s_nop 1` (VALU write -> DPP read of the same VGPR needs two wait states; inside a statement every instruction reads a
// register written at least a statement earlier).
#define S3D_SEG_STEP4(CTRL, a, b, c, d, m)                                                                  \
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %4 " CTRL "\n\tv_fmac_f32_dpp %1, %1, %4 " CTRL          \
                 "\n\tv_fmac_f32_dpp %2, %2, %4 " CTRL "\n\tv_fmac_f32_dpp %3, %3, %4 " CTRL                  \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m))
#define S3D_SEG_FLAG(CTRL, f) asm volatile("s_nop 1\n\tv_or_b32_dpp %0, %0, %0 " CTRL : "+v"(f))
#define S3D_SEG_STEP(CTRL)                                                                   \
    {                                                                                        \
        if (__ballot(f == 0u) == 0ull) return;                                               \
        const float m = f == 0u ? 1.0f : 0.0f;                                               \
        _Pragma("unroll") for (uint32_t i = 0; i < N; i += 4) S3D_SEG_STEP4(CTRL, v[i], v[i + 1], v[i + 2], v[i + 3], m); \
        S3D_SEG_FLAG(CTRL, f);                                                               \
    }
template <uint32_t N>
__device__ __forceinline__ void seg_scan_wave(float (&v)[N], bool head, uint32_t lane) {
    static_assert(N % 4 == 0, "four values per asm statement");
    (void)lane;
    uint32_t f = head ? 1u : 0u;
    S3D_SEG_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
    S3D_SEG_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
    S3D_SEG_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
    S3D_SEG_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
    S3D_SEG_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")  // lane 15 of rows 0 / 2 -> rows 1 / 3
    S3D_SEG_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")  // lane 31 -> rows 2, 3
}
#undef S3D_SEG_STEP
#undef S3D_SEG_FLAG
#undef S3D_SEG_STEP4

// the 32-bit value word of a record: the C products rounded to T (fp16 C = 2: one v_cvt_pk_f16_f32, round to nearest even
// like the two single conversions)
template <typename T, uint32_t C>
__device__ __forceinline__ uint32_t pack_record(const float* v) {
    uint32_t bits;
    if constexpr (sizeof(T) == 2 && C == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        f2 f;
        f[0] = v[0]; f[1] = v[1];
        const h2 h = __builtin_convertvector(f, h2);
        __builtin_memcpy(&bits, &h, 4);
    } else {
        T pr[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) pr[c] = Acc<T>::from_f(v[c]);
        __builtin_memcpy(&bits, pr, 4);
    }
    return bits;
}

// inclusive wave64 prefix sum by DPP (row_shr inside the 16-lane rows, then the two row_bcast carries)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v, uint32_t lane) {
    const uint32_t rl = lane & 15u, row = lane >> 4;
    uint32_t u;
    u = dpp_take<0x111, 0xF>(v); v += rl >= 1 ? u : 0u;
    u = dpp_take<0x112, 0xF>(v); v += rl >= 2 ? u : 0u;
    u = dpp_take<0x114, 0xF>(v); v += rl >= 4 ? u : 0u;
    u = dpp_take<0x118, 0xF>(v); v += rl >= 8 ? u : 0u;
    u = dpp_take<0x142, 0xA>(v); v += (row & 1u) ? u : 0u;
    u = dpp_take<0x143, 0xC>(v); v += row >= 2u ? u : 0u;
    return v;
}

// Bucket (level, slice, sub) = `cap` records: keys at gkeys[((level_in_pass * smax + slice) * kBin3Sub + sub) * cap], values
// at the same index of gvals; cursor[((level * smax + slice) * kBin3Sub + sub) * kCurStride] counts the records reserved in
// it (it may run past cap: the excess lives in the spill regions).
// Phases of a workgroup (profiled with -DS3D_BIN3_PROF, tools/debug/prof_bwd.py): loads in flight across the first
// (LDS-only) barrier -> products + wave scan -> LDS rank atomics, arrival counter -> the LAST wave to arrive scans the
// slice counts and puts the bucket reservations in flight -> barrier -> staging (sorted by slice) while the reservations
// return -> barrier -> copy-out, four records per lane in flight.
template <typename T, uint32_t D, uint32_t C, bool FIXED24, uint32_t P>
__global__ void __launch_bounds__(P) k_bin_scatter6(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                    const int32_t* __restrict__ offsets, uint32_t B, uint32_t level0,
                                                    LevelScales scales, uint32_t* __restrict__ hdr, uint32_t* __restrict__ cursor,
                                                    uint32_t* __restrict__ ovn, uint2* __restrict__ ovl, uint32_t smax,
                                                    uint32_t nchunks, uint32_t cap, uint16_t* __restrict__ gkeys,
                                                    uint32_t* __restrict__ gvals, uint16_t* __restrict__ skeys,
                                                    uint32_t* __restrict__ svals, uint32_t gridtype, bool align_corners,
                                                    uint32_t interp) {
    using V = typename FeatVec<T, C>::type;
    static_assert(sizeof(V) == 4, "records carry one 32-bit value word");
    constexpr uint32_t K = 1u << D;
    constexpr uint32_t NWV = P / 64;
    static_assert(P * K < 65536 && P % 64 == 0, "positions inside a chunk are 16-bit fields");
    static_assert(kBinAccBytes / (8 * C) <= 65536, "row-in-slice is a 16-bit key");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (tables sized by the call's largest slice count, not by kBinMaxSlices: at 64 slices the workgroup needs 33 KB instead of
    //  40 KB + 12 B — which was 52 bytes too much for a FOURTH workgroup per CU)
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem_raw);           // [smax] records of the chunk per slice
    uint32_t* lo = cnt + smax;                                        // [smax] run start in the chunk's sorted order
    uint2* tab = reinterpret_cast<uint2*>(lo + smax);                 // [smax] {bucket position of the run - run start, run start | records that fit << 16}
    uint2* stage = tab + smax;                                        // [P * K] {value, slice << 16 | row-in-slice}
    __shared__ uint32_t total_s, arrived, spilled_s;

    const uint32_t lip = blockIdx.y;  // level inside the pass
    const uint32_t level = level0 + lip, chunk = blockIdx.x;
    const uint32_t wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    (void)wg_lin;
    S3D_STAMP(0, wg_lin, 0);
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t Bv = valid_rows(B, scales.n_valid);
    if (chunk * P >= Bv) return;  // (uniform) chunk entirely in the absent tail of a padded batch
    if constexpr (!FIXED24) {
        const float amax = __uint_as_float(hdr[level]);
        if (!(amax > 0.0f) || amax == INFINITY) return;  // zero / non-finite levels carry no records (uniform exit)
    }
    const uint32_t b = chunk * P + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    float x[D];
    T g[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) g[c] = Acc<T>::zero();
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = 0.0f;
    const bool in = b < Bv;
    const uint32_t S = bin_slices(hashmap_size, C);
    const uint32_t sshift = 31 - __clz(S);
    const uint32_t sub = xcc_id() & (kBin3Sub - 1);
    const float lscale = scales.v[level];
    LevelIndex<D> li;
    li.init(gridtype, align_corners, hashmap_size, (uint32_t)ceilf(lscale) + 1);
    for (uint32_t s = threadIdx.x; s < S; s += P) cnt[s] = 0;
    if (threadIdx.x == 0) { arrived = 0; spilled_s = 0; }
    lds_barrier();  // counters cleared before anybody ranks a record (all waves are at their start: the barrier is free)
    if (in) {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
        load_feat<T, C>(grad + ((size_t)level * B + b) * C, g);
    }

    bool active = in;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {  // load_point()'s normalisation and range test
        if (scales.bound != 0.0f) x[d] = (x[d] + scales.bound) * scales.inv_2bound;
        if (x[d] < 0 || x[d] > 1) active = false;
    }
    bool nz = false;
    (void)absmax_feat<T, C>(g, nz);
    active = active && nz;  // samples behind a ray's termination (and padding rows) carry exact zeros
    S3D_STAMP(0, wg_lin, 1);
    float pos[D], pd[D];
    uint32_t pg[D];
    locate<D>(x, lscale, align_corners, interp, pos, pd, pg);
    float v[K * C];
    float gf[C];
    bool gbad = false;  // a non-finite gradient on an active lane poisons the level (FIXED24)
    float gmax = 0.0f;  // (NaN-propagating: v_max would drop a NaN operand)
#pragma unroll
    for (uint32_t c = 0; c < C; c++) {
        gf[c] = active ? Acc<T>::to_f(g[c]) : 0.0f;  // (inactive lanes: zero products, whatever their position)
        gbad |= !(fabsf(gf[c]) <= 3.402823466e38f);
        gmax = (fabsf(gf[c]) > gmax || gf[c] != gf[c]) ? fabsf(gf[c]) : gmax;
    }
#pragma unroll
    for (uint32_t idx = 0; idx < K; idx++) {
        float w = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) w *= ((idx >> d) & 1u) ? pos[d] : 1 - pos[d];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) v[idx * C + c] = w * gf[c];
    }
    // run = maximal stretch of consecutive ACTIVE lanes of the wave in one cell.  An inactive lane (zero gradient, out of
    // range) ends the run before it: in training those are the samples behind a ray's termination, i.e. the tail of a ray,
    // and the next active lane belongs to another ray anyway — and head flags that depend on the previous lane only keep
    // the scan at log2(longest run) steps.
    // (every cross-lane read is executed by the whole wave: under a partial exec mask disabled lanes read as 0)
    const uint32_t pa = dpp_take<0x138, 0xF>(active ? 1u : 0u);  // wave_shr:1 (lane 0 keeps its own value: excluded below)
    bool cells_equal = active && pa != 0u && lane > 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) cells_equal &= (dpp_take<0x138, 0xF>(pg[d]) == pg[d]);
    bool same = cells_equal;  // continues the run of lane - 1
    // a wave in which (almost) nothing merges skips the scan: every active lane is its own run
    const unsigned long long smask = __ballot(same);
    if (__popcll(smask) < S3D_BIN3_MIN_MERGES) same = false;
    else seg_scan_wave<K * C>(v, !same, lane);
    const uint32_t next_same = dpp_take<0x130, 0xF>(same ? 1u : 0u);  // wave_shl:1
    const bool tail = active && (lane == 63u || next_same == 0u);  // the run's last lane holds its sums
    S3D_STAMP(0, wg_lin, 2);

    uint32_t key[K], rank[K], val[K];
    if constexpr (FIXED24) {
        // hdr[level]: 0 = every record of the level is below 64 in magnitude (the accumulate converts with one multiply and one
        // conversion), 1 = larger values present (general conversion), NaN pattern = a record is non-finite after rounding
        // to binary16 (|sum| >= 65520 or a non-finite gradient): the level is poisoned like a float sum would be.
        // One maximum per lane instead of a test per record; at most one atomic per wave.
        // (|w| <= 1 and a run has at most 64 lanes: a wave whose gradients all lie below 1 in magnitude cannot hold a record of
        //  64 or more — one maximum and one ballot instead of 2^D x C maxima; NaN fails the comparison and takes the full test)
        if (__ballot(!(gmax < 1.0f)) != 0ull) {
            float m = 0.0f;
#pragma unroll
            for (uint32_t i = 0; i < K * C; i++) m = fmaxf(m, fabsf(v[i]));
            const bool bad = (active && gbad) || (tail && m >= 65520.0f);
            const bool big = tail && m >= 64.0f;
            const unsigned long long anybad = __ballot(bad), anybig = __ballot(big);
            if ((anybad | anybig) && lane == 0) atomicMax(hdr + level, anybad ? 0x7fc00000u : 1u);
        }
    }
    if (tail) {
        uint32_t row[K];
        if (li.hashed && li.pow2) {
            // (level-uniform branch) hashed level of power-of-two size: the 2^D rows are XORs of 2 D terms, one mask each —
            // no select between the dense and the hashed rule, no division path
            uint32_t t0[D], t1[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                t0[d] = pg[d] * kPrimes[d];
                t1[d] = t0[d] + kPrimes[d];
            }
#pragma unroll
            for (uint32_t idx = 0; idx < K; idx++) {
                uint32_t r = (idx & 1u) ? t1[0] : t0[0];
#pragma unroll
                for (uint32_t d = 1; d < D; d++) r ^= ((idx >> d) & 1u) ? t1[d] : t0[d];
                row[idx] = r & li.mask;
            }
        } else {
            uint32_t lo_[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) lo_[d] = pg[d] * li.mul[d];
#pragma unroll
            for (uint32_t idx = 0; idx < K; idx++) row[idx] = li.row(lo_, idx);
        }
        const uint32_t hshift = sshift + kBinGroupLog;
#pragma unroll
        for (uint32_t idx = 0; idx < K; idx++) {
            const uint32_t slice = (row[idx] >> kBinGroupLog) & (S - 1);
            key[idx] = (slice << 16) | ((row[idx] >> hshift) << kBinGroupLog) | (row[idx] & (kBinGroup - 1u));
            rank[idx] = atomicAdd(&cnt[slice], 1u);
            val[idx] = pack_record<T, C>(&v[idx * C]);
        }
    }
    // the last wave to arrive (its arrival is ordered behind every wave's rank atomics) finds the run starts and puts the
    // bucket reservations in flight; everybody else goes straight to the barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    uint32_t order = 0;
    if (lane == 0) order = atomicAdd(&arrived, 1u);
    order = __builtin_amdgcn_readfirstlane(order);
    const bool last = order == NWV - 1;
    S3D_STAMP(0, wg_lin, 3);
    constexpr uint32_t kPer = kBinMaxSlices / 64;
    uint32_t at[kPer];
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) at[j] = 0;
    if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        uint32_t carry = 0;
        {  // slices 0..63 (every level has them): the atomic is unconditional — a branch around it would make hipcc wait for
           // the result at the join (`at` becomes a phi of arrays), and the point is to let it fly across the barrier
            const uint32_t c = cnt[lane];
            const uint32_t incl = wave_incl_scan_dpp(c, lane);
            lo[lane] = incl - c;
            carry = __shfl(incl, 63, 64);
            at[0] = atomicAdd(&cursor[(((size_t)level * smax + lane) * kBin3Sub + sub) * kCurStride], c);
        }
        if (S > 64) {  // (tables beyond 64 slices: the results are awaited here)
#pragma unroll
            for (uint32_t j = 1; j < kPer; j++) {
                if (j * 64 < S) {
                    const uint32_t sl = j * 64 + lane, c = cnt[sl];
                    const uint32_t incl = wave_incl_scan_dpp(c, lane);
                    lo[sl] = carry + incl - c;
                    carry += __shfl(incl, 63, 64);
                    if (c) at[j] = atomicAdd(&cursor[(((size_t)level * smax + sl) * kBin3Sub + sub) * kCurStride], c);
                }
            }
        }
        if (lane == 0) total_s = carry;
    }
    lds_barrier();  // run starts visible; the reservations are still on their way
    S3D_STAMP(0, wg_lin, 4);
    if (tail) {
#pragma unroll
        for (uint32_t idx = 0; idx < K; idx++) stage[lo[key[idx] >> 16] + rank[idx]] = make_uint2(val[idx], key[idx]);
    }
    if (last) {  // reservations back: what fits the bucket, where it goes, what is spilled
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            if (j * 64 < S) {
                const uint32_t sl = j * 64 + lane, c = cnt[sl], l0 = lo[sl], a = at[j];
                const uint32_t ft = a >= cap ? 0u : (cap - a < c ? cap - a : c);
                tab[sl] = make_uint2((uint32_t)((int32_t)((sl * kBin3Sub + sub) * cap + a) - (int32_t)l0), l0 | (ft << 16));
                if (ft < c) {
                    spilled_s = 1u;  // (benign race: every writer stores 1)
                    const uint32_t k = atomicAdd(&ovn[level * smax + sl], 1u);
                    ovl[((size_t)level * smax + sl) * nchunks + k] = make_uint2((chunk << 16) | (l0 + ft), c - ft);
                }
            }
        }
    }
    lds_barrier();
    S3D_STAMP(0, wg_lin, 5);
    const uint32_t total = total_s;
    // (uniform bases + 32-bit record indices: the stores take the scalar-base form, no 64-bit address arithmetic per record)
    uint16_t* const gk = gkeys + (size_t)lip * smax * kBin3Sub * cap;
    uint32_t* const gv = gvals + (size_t)lip * smax * kBin3Sub * cap;
    uint16_t* const sk = skeys + ((size_t)lip * nchunks + chunk) * (P * K);
    uint32_t* const sv = svals + ((size_t)lip * nchunks + chunk) * (P * K);
    constexpr uint32_t UC = 4;
    if (!spilled_s) {
        // every run fits its bucket (the rule; tested once per workgroup, not per record): position = sorted index + the
        // run's bucket offset.  Whole trips of UC x P records carry no per-record range test (the trip count is uniform);
        // the remainder goes one record per lane and trip.
        uint32_t base = 0;
        for (; base + UC * P <= total; base += UC * P) {
            uint2 r[UC];
            uint32_t gd[UC];
#pragma unroll
            for (uint32_t u = 0; u < UC; u++) r[u] = stage[base + threadIdx.x + u * P];
#pragma unroll
            for (uint32_t u = 0; u < UC; u++) gd[u] = tab[r[u].y >> 16].x;
#pragma unroll
            for (uint32_t u = 0; u < UC; u++) {
                const uint32_t p = base + threadIdx.x + u * P + gd[u];  // (mod 2^32: gd = bucket position - run start)
                gk[p] = (uint16_t)r[u].y;
                gv[p] = r[u].x;
            }
        }
        for (uint32_t i = base + threadIdx.x; i < total; i += P) {
            const uint2 r = stage[i];
            const uint32_t p = i + tab[r.y >> 16].x;
            gk[p] = (uint16_t)r.y;
            gv[p] = r.x;
        }
    } else {
        for (uint32_t i0 = threadIdx.x; i0 < total; i0 += UC * P) {
            uint2 r[UC], t[UC];
#pragma unroll
            for (uint32_t u = 0; u < UC; u++) {
                const uint32_t i = i0 + u * P;
                r[u] = stage[i < total ? i : 0];
            }
#pragma unroll
            for (uint32_t u = 0; u < UC; u++) t[u] = tab[r[u].y >> 16];
#pragma unroll
            for (uint32_t u = 0; u < UC; u++) {
                const uint32_t i = i0 + u * P;
                if (i < total) {
                    if (i - (t[u].y & 0xffffu) < (t[u].y >> 16)) {
                        const uint32_t p = i + t[u].x;
                        gk[p] = (uint16_t)r[u].y;
                        gv[p] = r[u].x;
                    } else {
                        sk[i] = (uint16_t)r[u].y;
                        sv[i] = r[u].x;
                    }
                }
            }
        }
    }
    S3D_STAMP(0, wg_lin, 6);
#ifdef S3D_BIN3_PROF
    __builtin_amdgcn_s_waitcnt(0);
    S3D_STAMP(0, wg_lin, 7);
#endif
}

// Accumulate, persistent: one workgroup per CU (the 128 KiB of accumulators fill its LDS) walks the (level, slice) items
// w, w + G, w + 2G, ... of the pass.  Between two items nothing is re-initialised: the write-out clears every accumulator
// behind its read, so only the first item pays the 128 KiB zero fill; the table rows a lane will add to are requested
// before the streaming phase.  Profiled (tools/debug/prof_bwd.py): the streaming phase runs at the LDS rate of 64-bit
// atomics (~4 per clock and CU, two per record) — that, not HBM, is this kernel's floor.
// The kernel also leaves the control block the way it found it (all zero): every item clears its cursors and its spill
// count after reading them, and the last workgroup to finish (ticket) clears the levels' header words — the caller's next
// call needs no clearing launch.
// ADAM (build extension, s3d_grid_encode_backward_adam): the table's parameter update applied where the row sums are —
// the write-out walks EVERY row of the item's slice (rows without records take g = 0: Adam moves them on their moments, as
// torch.optim.Adam on a dense gradient does) and runs the optimizer's own update arithmetic (s3d_adam.hpp) on the fp32 master
// row, its two moments and the fp16 copy the next forward reads.  The gradient table is neither read nor written.  The row's
// gradient is the exact sum rounded to binary16 — the value the unfused path stores and the optimizer reads back, so both routes
// give the same bits — except where that rounding would overflow: then the fp32 sum is used (the unfused path raises
// GradScaler's flag there; this kernel cannot, rows of other items have been written by then).  The skip decision of the step
// is taken BEFORE the first row is touched: the flag other producers of the step raised (MLP reduce, ...) and the poison words
// of every level of this call (a non-finite dL/dy seen by the scatter).
struct GridAdam {
    float* p;      // fp32 master table [rows, C]
    float* m;      // exp_avg
    float* v;      // exp_avg_sq
    __half* ph;    // fp16 copy of p (or nullptr)
    float lr, beta1, beta2, eps;
    const float* step;        // device: step count before this update
    const float* grad_scale;  // device: loss scale (or nullptr)
    const float* lr_scale;    // device: schedule factor (or nullptr)
};
template <typename T, uint32_t D, uint32_t C, bool FIXED24, uint32_t P, bool ADAM = false>
__global__ void __launch_bounds__(kBinAccThreads) k_bin_accumulate6(const uint16_t* __restrict__ gkeys, const uint32_t* __restrict__ gvals,
                                                                   const uint16_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                                                                   const int32_t* __restrict__ offsets, T* __restrict__ grad_grid,
                                                                   uint32_t B, uint32_t level0, uint32_t nl, uint32_t* __restrict__ hdr,
                                                                   uint32_t* __restrict__ done, uint32_t* __restrict__ cursor,
                                                                   uint32_t* __restrict__ ovn, const uint2* __restrict__ ovl,
                                                                   uint32_t smax, uint32_t nchunks, uint32_t cap,
                                                                   float* __restrict__ found_inf, const GridAdam ad) {
    using V = typename FeatVec<T, C>::type;
    static_assert(!ADAM || (FIXED24 && sizeof(T) == 2 && C == 2), "the fused update is built for the fp16 C = 2 tables of the -O configs");
    static_assert(sizeof(V) == 4, "records carry one 32-bit value word");
    constexpr uint32_t K = 1u << D;
    constexpr uint32_t NS = kBin3Sub;
    constexpr uint32_t NW = kBinAccThreads / 64;
    static_assert(NW % NS == 0, "whole waves per sub-bucket");
    constexpr uint32_t STEP = (NW / NS) * 64;  // quads of one sub-bucket taken per step by its waves
    constexpr uint32_t W = 8;                  // table rows per lane and round of the write-out
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem_raw);  // [C][local_rows]
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t sj = wave % NS;  // sub-bucket streamed by this wave: count, base and trip count are wave-uniform
    const uint32_t n_items = smax * nl;
    bool overflow = false;  // a finite sum that leaves the range of T (fp16: |v| > 65504) — what GradScaler looks for
    bool dirty = true;      // accumulators not known to be zero (first item)
    uint32_t it_no = 0;
    bool ad_skip = false;
    AdamCoef ac{};
    if constexpr (ADAM) {
        ad_skip = found_inf && *found_inf != 0.0f;
        bool pois = false;
        for (uint32_t l = 0; l < nl; l++) pois |= hdr[level0 + l] >= 0x7f800000u;  // (uniform: scalar loads)
        if (pois && found_inf && threadIdx.x == 0) *found_inf = 1.0f;              // (benign race: everyone writes 1)
        ad_skip |= pois;
        ac = adam_coef(ad.lr, ad.beta1, ad.beta2, ad.eps, 0.0f, ad.step, ad.grad_scale, ad.lr_scale);
    }
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x, it_no++) {
        const uint32_t lip = item / smax, slice = item - lip * smax;  // level inside the pass
        const uint32_t level = level0 + lip;
        const uint32_t wg_lin = item;
        (void)wg_lin;
        // every global word of the prologue is requested before the first one is tested
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t rows = (uint32_t)offsets[level + 1] - off;
        const uint32_t hbits = hdr[level];
        uint32_t* cur_w = cursor + ((size_t)level * smax + slice) * NS * kCurStride;
        uint32_t reserved = 0;
#pragma unroll
        for (uint32_t j = 0; j < NS; j++) reserved |= cur_w[j * kCurStride];
        uint32_t cj = cur_w[sj * kCurStride];
        const uint32_t nspill = ovn[level * smax + slice];
        S3D_STAMP(1, wg_lin, 0);
        const uint32_t S = bin_slices(rows, C);
        if (slice >= S) continue;  // (no such slice: its control words were never touched)
        const uint32_t local_rows = bin_local_rows(rows, S);
        auto row_of_local = [&](uint32_t local) { return ((local / kBinGroup) * S + slice) * kBinGroup + local % kBinGroup; };
        T* table = grad_grid + (size_t)off * C;
        const float amax = __uint_as_float(hbits);
        const bool poisoned = FIXED24 ? hbits >= 0x7f800000u : (amax != amax || amax == INFINITY);
        const bool small = FIXED24 && hbits == 0u;  // every record of the level below 64: v * 2^24 fits 31 bits
        // the control words of this item go back to zero (all of its readers — this workgroup's lanes — hold them in registers;
        // the barrier below, or the end of the kernel, orders the stores behind the loads)
        const bool touched = reserved != 0u;  // (a spilled run implies a reservation)
        if (touched) {
            __syncthreads();
            if (threadIdx.x < NS) cur_w[threadIdx.x * kCurStride] = 0u;
            if (threadIdx.x == NS) ovn[level * smax + slice] = 0u;
        }
        if constexpr (ADAM) {
            if (ad_skip) continue;  // skipped step: nothing is updated (the control words above are clean again)
            const bool stream_it = touched;
            const uint32_t cj_a = cj < cap ? cj : cap;
            const uint32_t nquads_a = (cj_a + 3) / 4;
            const size_t base_a = (((size_t)lip * smax + slice) * NS + sj) * cap;
            auto add_a = [&](auto small_c, uint32_t key, uint32_t bits) {
                T pr[C];
                __builtin_memcpy(pr, &bits, 4);
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    const float vv = Acc<T>::to_f(pr[c]);
                    long long q;
                    if constexpr (decltype(small_c)::value) q = (long long)(int)(vv * 16777216.0f);
                    else q = fixed24_exact(vv);
                    atomicAdd(&acc[c * local_rows + key], (unsigned long long)q);
                }
            };
            // the optimizer state of EVERY row of the slice (8 rows per lane: 48 registers) is requested before the records are
            // streamed: the LDS-bound streaming phase hides the 24 B per parameter of reads, the write-out is arithmetic + stores
            typedef float f2v __attribute__((ext_vector_type(2)));
            float* Pm = ad.p + (size_t)off * C;
            float* Mm = ad.m + (size_t)off * C;
            float* Vm = ad.v + (size_t)off * C;
            __half* Hm = ad.ph ? ad.ph + (size_t)off * C : nullptr;
            constexpr uint32_t WA = kBinAccBytes / (8 * C) / kBinAccThreads;  // rows per lane: the whole slice in one round
            static_assert(WA * kBinAccThreads * 8 * C == kBinAccBytes && WA <= 8, "a slice is WA rows per lane");
            f2v pv[WA], mv[WA], vv[WA];
            uint32_t rowi[WA];
#pragma unroll
            for (uint32_t w = 0; w < WA; w++) {
                const uint32_t rr = threadIdx.x + w * kBinAccThreads;
                const uint32_t row = rr < local_rows ? row_of_local(rr) : 0xffffffffu;
                rowi[w] = row < rows ? row : 0xffffffffu;
            }
            auto request_state = [&](uint32_t w) {  // (w is a compile-time index at every call site: the arrays stay in registers)
                if (rowi[w] != 0xffffffffu) {
                    pv[w] = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(Pm + (size_t)rowi[w] * C));
                    mv[w] = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(Mm + (size_t)rowi[w] * C));
                    vv[w] = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(Vm + (size_t)rowi[w] * C));
                }
            };
            // the optimizer state of EVERY row of the slice (8 rows per lane: 48 registers) is requested before the records are
            // streamed.  [Measured, profiles/r11_grid_backward.md: requests riding behind each trip's records instead (vector
            // loads return in order, so state in front of the records holds the first trip back) need the trips unrolled and
            // spill ~50 registers at 1,024 threads: 148 - 152 us against 117 - 121 us for this arrangement at 1.1e5 points.]
#pragma unroll
            for (uint32_t w = 0; w < WA; w++) request_state(w);
            if (stream_it) {
                if (dirty) {
                    for (uint32_t i = threadIdx.x; i < kBinAccBytes / 8; i += kBinAccThreads) acc[i] = 0ull;
                    dirty = false;
                }
                lds_barrier();  // accumulators clear (this item's fill, or the previous item's write-out); the state loads stay in flight
                auto stream_a = [&](auto small_c) {
                    const uint32_t q0a = (wave / NS) * 64 + (threadIdx.x & 63u);
                    for (uint32_t qi = q0a; qi < nquads_a; qi += STEP) {
                        const uint2 k = *reinterpret_cast<const uint2*>(gkeys + base_a + 4 * (size_t)qi);
                        const uint4 vq = *reinterpret_cast<const uint4*>(gvals + base_a + 4 * (size_t)qi);
                        const uint32_t nq = cj_a - 4 * qi < 4u ? cj_a - 4 * qi : 4u;
                        add_a(small_c, k.x & 0xffffu, vq.x);
                        if (nq > 1) add_a(small_c, k.x >> 16, vq.y);
                        if (nq > 2) add_a(small_c, k.y & 0xffffu, vq.z);
                        if (nq > 3) add_a(small_c, k.y >> 16, vq.w);
                    }
                };
                if (small) stream_a(std::true_type{});
                else stream_a(std::false_type{});
                for (uint32_t k = threadIdx.x >> 6; k < nspill; k += kBinAccThreads / 64) {
                    const uint2 d = ovl[((size_t)level * smax + slice) * nchunks + k];
                    const size_t run = ((size_t)lip * nchunks + (d.x >> 16)) * (P * K) + (d.x & 0xffffu);
                    for (uint32_t j = threadIdx.x & 63u; j < d.y; j += 64) add_a(std::false_type{}, skeys[run + j], svals[run + j]);
                }
                lds_barrier();
            }
#pragma unroll
            for (uint32_t w = 0; w < WA; w++) {
                if (rowi[w] == 0xffffffffu) continue;
                const uint32_t rr = threadIdx.x + w * kBinAccThreads;
                float g[C];
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    g[c] = 0.0f;
                    if (stream_it) {
                        const long long q = (long long)acc[c * local_rows + rr];
                        if (q != 0) {
                            acc[c * local_rows + rr] = 0ull;  // cleared behind the read: the next item finds zeros
                            const float sum = fixed_to_float(q, (int)kFixedExp);
                            const float h = Acc<T>::to_f(Acc<T>::from_f(sum));  // the value the gradient table would hold
                            g[c] = fabsf(h) <= 65504.0f ? h : sum;
                        }
                    }
                }
                float pp[C] = {pv[w].x, pv[w].y}, mm[C] = {mv[w].x, mv[w].y}, vq[C] = {vv[w].x, vv[w].y};
#pragma unroll
                for (uint32_t c = 0; c < C; c++) adam_update(ac, g[c], mm[c], vq[c], pp[c]);
                __builtin_nontemporal_store(f2v{mm[0], mm[1]}, reinterpret_cast<f2v*>(Mm + (size_t)rowi[w] * C));
                __builtin_nontemporal_store(f2v{vq[0], vq[1]}, reinterpret_cast<f2v*>(Vm + (size_t)rowi[w] * C));
                __builtin_nontemporal_store(f2v{pp[0], pp[1]}, reinterpret_cast<f2v*>(Pm + (size_t)rowi[w] * C));
                if (Hm) *reinterpret_cast<__half2*>(Hm + (size_t)rowi[w] * C) = __floats2half2_rn(pp[0], pp[1]);
            }
            continue;
        }
        if (poisoned) {
            for (uint32_t i = threadIdx.x; i < local_rows * C; i += kBinAccThreads) {
                const uint32_t row = row_of_local(i / C);
                if (row < rows) table[(size_t)row * C + i % C] = Acc<T>::from_f(NAN);
            }
            if (found_inf && threadIdx.x == 0) *found_inf = 1.0f;  // (benign race: everyone writes 1)
            continue;
        }
        if (!FIXED24 && !(amax > 0.0f)) continue;
        if (!touched) continue;
        cj = cj < cap ? cj : cap;
        const uint32_t nquads = (cj + 3) / 4;  // groups of four records in this wave's sub-bucket
        int kexp = (int)kFixedExp;
        if constexpr (!FIXED24) {
            int e;
            (void)frexpf(amax, &e);
            kexp = 62 - e - (int)(32 - __clz(B)) - (int)D;
        }
        const size_t base = (((size_t)lip * smax + slice) * NS + sj) * cap;  // cap is a multiple of 64: 8- / 16-byte aligned
        auto add_t = [&](auto small_c, uint32_t key, uint32_t bits) {
            T pr[C];
            __builtin_memcpy(pr, &bits, 4);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                const float v = Acc<T>::to_f(pr[c]);
                long long q;
                if constexpr (decltype(small_c)::value) q = (long long)(int)(v * 16777216.0f);  // exact: a multiple of 2^-24 below 64
                else q = FIXED24 ? fixed24_exact(v) : to_fixed64(v, kexp);
                atomicAdd(&acc[c * local_rows + key], (unsigned long long)q);
            }
        };
        auto add = [&](uint32_t key, uint32_t bits) { add_t(std::false_type{}, key, bits); };
        struct Quad { uint2 k; uint4 v; uint32_t n; };
        auto fetch = [&](uint32_t qi) {
            Quad q;
            q.n = 0;
            q.k = make_uint2(0u, 0u);
            q.v = make_uint4(0u, 0u, 0u, 0u);
            if (qi < nquads) {
                q.k = *reinterpret_cast<const uint2*>(gkeys + base + 4 * (size_t)qi);
                q.v = *reinterpret_cast<const uint4*>(gvals + base + 4 * (size_t)qi);
                q.n = cj - 4 * qi < 4u ? cj - 4 * qi : 4u;
            }
            return q;
        };
        auto consume = [&](auto small_c, const Quad& q) {
            if (q.n == 4) {  // (all but the last group of a sub-bucket)
                add_t(small_c, q.k.x & 0xffffu, q.v.x);
                add_t(small_c, q.k.x >> 16, q.v.y);
                add_t(small_c, q.k.y & 0xffffu, q.v.z);
                add_t(small_c, q.k.y >> 16, q.v.w);
            } else {
                if (q.n > 0) add_t(small_c, q.k.x & 0xffffu, q.v.x);
                if (q.n > 1) add_t(small_c, q.k.x >> 16, q.v.y);
                if (q.n > 2) add_t(small_c, q.k.y & 0xffffu, q.v.z);
            }
        };
        S3D_STAMP(1, wg_lin, 1);
        // the first batch of records and the table rows of the write-out's first round are in flight while the accumulators
        // are cleared (first item only)
#ifndef S3D_BIN3_ACC_U  // groups of four records in flight per lane (plus as many prefetched).  Measured (r08, 2.6e5 ray-ordered points): 1: 104.8 us, 2: 107-112, 3: 122.8, 4: 155.4 (register spills) for the whole backward
#define S3D_BIN3_ACC_U 1
#endif
        constexpr uint32_t U = S3D_BIN3_ACC_U;
        const uint32_t q0 = (wave / NS) * 64 + (threadIdx.x & 63u);
        Quad cur[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) cur[u] = fetch(q0 + u * STEP);
        V old0[W];
#pragma unroll
        for (uint32_t w = 0; w < W; w++) {
            const uint32_t rr = threadIdx.x + w * kBinAccThreads;
            uint32_t zero = 0;
            __builtin_memcpy(&old0[w], &zero, 4);
            if (rr < local_rows) {
                const uint32_t row = row_of_local(rr);
                if (row < rows) old0[w] = *reinterpret_cast<const V*>(table + (size_t)row * C);
            }
        }
        if (dirty) {
            for (uint32_t i = threadIdx.x; i < kBinAccBytes / 8; i += kBinAccThreads) acc[i] = 0ull;
            dirty = false;
        }
        __syncthreads();  // accumulators clear (this item's fill, or the previous item's write-out)
        S3D_STAMP(1, wg_lin, 2);
        auto stream = [&](auto small_c) {
            for (uint32_t b0 = 0; b0 < nquads; b0 += U * STEP) {  // (wave-uniform trip count)
                Quad nx[U];
#pragma unroll
                for (uint32_t u = 0; u < U; u++) nx[u] = fetch(b0 + q0 + (U + u) * STEP);
#pragma unroll
                for (uint32_t u = 0; u < U; u++) consume(small_c, cur[u]);
#pragma unroll
                for (uint32_t u = 0; u < U; u++) cur[u] = nx[u];
            }
        };
        if (small) stream(std::true_type{});
        else stream(std::false_type{});
        S3D_STAMP(1, wg_lin, 3);
        // spilled runs (clustered samples only): one wave per descriptor
        for (uint32_t k = threadIdx.x >> 6; k < nspill; k += kBinAccThreads / 64) {
            const uint2 d = ovl[((size_t)level * smax + slice) * nchunks + k];
            const size_t run = ((size_t)lip * nchunks + (d.x >> 16)) * (P * K) + (d.x & 0xffffu);
            for (uint32_t j = threadIdx.x & 63u; j < d.y; j += 64) add(skeys[run + j], svals[run + j]);
        }
        __syncthreads();
        S3D_STAMP(1, wg_lin, 4);
        // write-out in rounds of WR rows per lane; the first W / WR rounds use the rows requested before the streaming phase
        constexpr uint32_t WR = 4;
        static_assert(W % WR == 0, "prefetched rows are consumed in whole rounds");
        uint32_t round = 0;
        for (uint32_t r0 = threadIdx.x; r0 < local_rows; r0 += WR * kBinAccThreads, round++) {
            long long q[WR][C];
            V old[WR];
            bool nz[WR];
#pragma unroll
            for (uint32_t w = 0; w < WR; w++) {
                const uint32_t rr = r0 + w * kBinAccThreads;
                nz[w] = false;
                if (rr < local_rows) {
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) {
                        q[w][c] = (long long)acc[c * local_rows + rr];
                        nz[w] |= (q[w][c] != 0);
                    }
                }
            }
#pragma unroll
            for (uint32_t w = 0; w < WR; w++) {  // clear behind the read: the next item finds zeros
                const uint32_t rr = r0 + w * kBinAccThreads;
                if (nz[w]) {
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) acc[c * local_rows + rr] = 0ull;
                }
            }
            if (round < W / WR) {
#pragma unroll
                for (uint32_t k = 0; k < W / WR; k++)
                    if (round == k) {
#pragma unroll
                        for (uint32_t w = 0; w < WR; w++) old[w] = old0[k * WR + w];
                    }
            } else {
#pragma unroll
                for (uint32_t w = 0; w < WR; w++)
                    if (nz[w]) old[w] = *reinterpret_cast<const V*>(table + (size_t)row_of_local(r0 + w * kBinAccThreads) * C);
            }
            // (this phase is bound by vector-instruction issue: a wave whose sums all fit 32 bits — the rule: |sum| < 128 at the
            //  fixed scale 2^24 — converts them with one cvt + one ldexp each instead of the two-limb route; same rounding)
            bool fits = true;
#pragma unroll
            for (uint32_t w = 0; w < WR; w++)
                if (nz[w]) {
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) fits &= (q[w][c] == (long long)(int)q[w][c]);
                }
            const bool narrow = __ballot(!fits) == 0ull;
#pragma unroll
            for (uint32_t w = 0; w < WR; w++) {
                if (nz[w]) {
                    T o[C];
                    __builtin_memcpy(o, &old[w], sizeof(V));
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) {
                        const float sum = narrow ? ldexpf((float)(int)q[w][c], -kexp) : fixed_to_float(q[w][c], kexp);
                        o[c] = Acc<T>::from_f(Acc<T>::to_f(o[c]) + sum);
                        overflow |= !(fabsf(Acc<T>::to_f(o[c])) <= 3.402823466e38f);
                    }
                    store_feat<T, C>(table + (size_t)row_of_local(r0 + w * kBinAccThreads) * C, o);
                }
            }
        }
        S3D_STAMP(1, wg_lin, 5);
    }
    if (found_inf && overflow) *found_inf = 1.0f;
    // the last workgroup to get here clears the header words of the pass's levels and the ticket itself (every workgroup has
    // read its header words before it takes a ticket)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(done, 1u);
        if (t == gridDim.x - 1) {
            for (uint32_t l = 0; l < nl; l++) hdr[level0 + l] = 0u;
            *done = 0u;
        }
    }
}

// s3d_grid_encode_backward(found_inf): the paths that do not report while they accumulate check the table afterwards
template <typename T>
__global__ void __launch_bounds__(256) k_table_nonfinite(const T* __restrict__ table, const int32_t* __restrict__ offsets, uint32_t L,
                                                         uint32_t C, float* __restrict__ found_inf) {
    const size_t n = (size_t)offsets[L] * C;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        bad |= !(fabsf(Acc<T>::to_f(table[i])) <= 3.402823466e38f);
    if (bad) *found_inf = 1.0f;
}
// found_inf of the entry point being served on this thread, and whether the path taken has reported into it
static thread_local float* t_found_inf = nullptr;
static thread_local bool t_reported = false;
static thread_local const struct GridAdam* t_adam = nullptr;  // s3d_grid_encode_backward_adam: the update to apply in the accumulate
static thread_local bool t_adam_applied = false;

// gridencoder.cu:340-366
template <typename T, uint32_t D, uint32_t C>
__global__ void k_grid_input_backward(const T* __restrict__ grad, const T* __restrict__ dy_dx, T* __restrict__ grad_inputs,
                                      uint32_t B, uint32_t L, const int32_t* __restrict__ n_valid) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= valid_rows(B, n_valid) * D) return;
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
    if (load_point<D>(inputs, b, scales, x)) return;
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

void host_scales(uint32_t L, float S, uint32_t H, LevelScales& out, float bound = 0.0f, const int32_t* n_valid = nullptr) {
    out.n_valid = n_valid;
    out.live = nullptr;
    out.live_stride = 0;
    out.bound = bound;
    out.inv_2bound = bound != 0.0f ? 1.0f / (2.0f * bound) : 0.0f;
    for (uint32_t l = 0; l < kMaxLevels; l++) out.v[l] = 0.0f;
    for (uint32_t l = 0; l < L; l++) out.v[l] = fmaf(exp2f((float)l * S), (float)H, -1.0f);
}

inline uint32_t xcd_grid(uint32_t B) { return kXcds * div_up<uint32_t>(B, kFwdBlock); }

// Level -> XCD plan of the lane-pair forward.  The kernel is bound by the L2's request rate, and every XCD has its own L2:
// with level l served by XCD l % 8 alone (the round-1 mapping) the XCDs finish at different times.  Measured on ray-ordered
// points (tools/fwd_levels.py, one level alone on its home XCD, B = 2^18, Lego configuration; profiles/r09_grid_forward_plan.md):
// levels 0 - 5 (resolution <= 81) cost 10 - 11 us each whatever their table, from there the cost grows with the resolution —
// a marching step crosses more and more cells, the 64 lanes of a gather share fewer and fewer lines — to 36 us from resolution
// ~700 on.  Home mapping: XCD 0 serves levels 0 + 8 = 28 us, XCD 7 levels 7 + 15 = 51 us; the launch ends with XCD 7.
// Here the work is cut in (level, chunk residue) slices — chunk = 256 points, residue = chunk % 16 — that start on XCD l % 8
// and are moved from the most to the least loaded XCD until nothing improves.  What may move: only home slices, and a
// receiver never holds more than two LARGE tables ((res + 1)^3 >= 2^18 rows: 2 MiB of fp16 pairs at the default hash size, an
// L2 is 4 MiB); preferred are levels the receiver already serves, then small (dense) levels, then the cheapest (most coherent)
// level.  Cost model fitted to the measurement: 0.29 up to resolution 81, linear to 1 at resolution ~700.  (Uniformly random
// points cost ~39 us on every level from 3 on, i.e. the home mapping is balanced for them and this plan costs them ~10 %; every
// product caller presents ray-, Morton- or lattice-ordered points.)  Placement only: every point is computed by the same
// instructions as before — results are bit-identical (tests/test_gpu_gridencoder.py runs against the oracle unchanged).
inline FwdPlan balance_forward_plan(uint32_t L, const LevelScales& sc, bool balance) {
    FwdPlan p;
    memset(&p, 0, sizeof(p));
    float cost[kMaxLevels], load[kXcds] = {0};
    bool big[kMaxLevels];
    uint8_t owner[kMaxLevels][kFwdResidues];
    for (uint32_t l = 0; l < L; l++) {
        const float res = ceilf(sc.v[l]) + 1.0f;
        const float c = 0.29f + 0.71f * (res - 81.0f) / 620.0f;
        cost[l] = (c < 0.29f ? 0.29f : (c > 1.0f ? 1.0f : c)) / (float)kFwdResidues;  // per slice
        big[l] = (res + 1.0f) * (res + 1.0f) * (res + 1.0f) >= 262144.0f;
        for (uint32_t k = 0; k < kFwdResidues; k++) owner[l][k] = (uint8_t)(l % kXcds);
        load[l % kXcds] += cost[l] * kFwdResidues;
    }
    auto serves = [&](uint32_t x, uint32_t l) {
        for (uint32_t k = 0; k < kFwdResidues; k++) if (owner[l][k] == x) return true;
        return false;
    };
    for (int it = 0; balance && it < 1024; it++) {
        uint32_t hi = 0;
        for (uint32_t x = 1; x < kXcds; x++) if (load[x] > load[hi]) hi = x;
        uint32_t order[kXcds];
        for (uint32_t x = 0; x < kXcds; x++) order[x] = x;
        for (uint32_t i = 1; i < kXcds; i++)
            for (uint32_t j = i; j > 0 && load[order[j]] < load[order[j - 1]]; j--) { const uint32_t t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
        bool moved = false;
        for (uint32_t oi = 0; oi < kXcds && !moved; oi++) {  // receivers from the least loaded up
            const uint32_t lo = order[oi];
            if (lo == hi) break;
            uint32_t n_big = 0;
            for (uint32_t l = 0; l < L; l++) n_big += (big[l] && serves(lo, l)) ? 1u : 0u;
            int best_l = -1, best_k = -1;
            float best_key = 0.0f;
            for (uint32_t l = 0; l < L; l++) {
                if (l % kXcds != hi) continue;  // (only home slices travel, and only once)
                if (load[hi] - cost[l] < load[lo] + cost[l]) continue;
                int k_own = -1;
                for (uint32_t k = 0; k < kFwdResidues; k++) if (owner[l][k] == hi) k_own = (int)k;
                if (k_own < 0) continue;
                const bool has = serves(lo, l);
                if (big[l] && !has && n_big >= 2u) continue;
                const float key = (has ? 0.0f : 100.0f) + (big[l] ? 10.0f : 0.0f) + cost[l];  // smaller = preferred
                if (best_l < 0 || key < best_key) { best_l = (int)l; best_k = k_own; best_key = key; }
            }
            if (best_l < 0) continue;
            owner[best_l][best_k] = (uint8_t)lo;
            load[hi] -= cost[best_l];
            load[lo] += cost[best_l];
            moved = true;
        }
        if (!moved) break;
    }
#ifdef S3D_FWD_PLAN_ENV  // calibration builds only (tools/fwd_levels.py): ONE level, all of it on its home XCD
    if (const char* only = getenv("S3D_FWD_ONLY_LEVEL")) {
        const uint32_t l = (uint32_t)atoi(only);
        for (uint32_t k = 0; k < kFwdResidues; k++) p.mask[l % kXcds][k] = l < L ? 1u << l : 0u;
        return p;
    }
#endif
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t k = 0; k < kFwdResidues; k++) p.mask[owner[l][k]][k] |= 1u << l;
    return p;
}

template <typename T, uint32_t D>
int launch_forward(const float* inputs, const T* emb, const int32_t* offsets, T* outputs, uint32_t B, uint32_t C,
                   uint32_t L, const LevelScales& sc, T* dy_dx, uint32_t gridtype, bool ac, uint32_t interp,
                   hipStream_t st, const T* emb_b = nullptr, T* outputs_b = nullptr) {
    // (chunk slots of the lane-pair kernel: the inference loop's launches — `live` rows, all of them filled — keep one
    //  workgroup per chunk; everything else walks its chunks from 2,048 slots per XCD, i.e. unchanged up to 2^19 rows)
    const uint32_t chunks = div_up<uint32_t>(B, kFwdBlock);
    // S3D_FWD_SLOTS=<n> (experiments, profiles/r11_grid_isolated.md): a persistent queue of n chunk slots per XCD for EVERY launch
    // — n = 32 CUs x resident workgroups is "sized to residency" — instead of one workgroup per chunk
    static const uint32_t forced_slots = [] { const char* e = getenv("S3D_FWD_SLOTS"); return e ? (uint32_t)atoi(e) : 0u; }();
    const uint32_t slots = forced_slots ? (forced_slots + kFwdResidues - 1) / kFwdResidues * kFwdResidues : kFwdMaxChunks;
    const bool strided = forced_slots ? chunks > slots
                                      : (!sc.live && sc.n_valid && chunks > kFwdMaxChunks);  // (a device-side row count: the extent may be mostly padding)
    const dim3 grid_pair(kXcds * (strided ? slots : chunks), emb_b ? 2u : 1u);
    const dim3 grid(xcd_grid(B), emb_b ? 2u : 1u), block(kFwdBlock);
    if (emb_b && (dy_dx || (sizeof(T) * C) % 4 != 0 || (sizeof(T) == 4 && C != 1 && C != 2 && C != 4 && C != 8) ||
                  (sizeof(T) == 2 && C != 2 && C != 4 && C != 8))) {
        set_error("grid_encode_forward_pair: served by the lane-pair kernel (whole 32-bit feature words, no input Jacobian)");
        return S3D_ERR_UNSUPPORTED;
    }
    FwdPlans plan;
    plan.p[0] = balance_forward_plan(L, sc, false);
    plan.p[1] = balance_forward_plan(L, sc, true);
    // lane-pair kernel whenever the feature vector is whole 32-bit words and no input Jacobian is asked for (the training and
    // inference paths of every configuration); k_grid_forward keeps fp16 C = 1 and the Jacobian
    if (!dy_dx && (sizeof(T) * C) % 4 == 0) {
        if constexpr (sizeof(T) == 4) {
            switch (C) {
                case 1: if (strided) hipLaunchKernelGGL((k_grid_forward_pair<T, D, 1, true>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    else hipLaunchKernelGGL((k_grid_forward_pair<T, D, 1>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    return check_launch("grid_encode_forward");
                default: break;
            }
        }
        switch (C) {
            case 2: if (strided) hipLaunchKernelGGL((k_grid_forward_pair<T, D, 2, true>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    else hipLaunchKernelGGL((k_grid_forward_pair<T, D, 2>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    return check_launch("grid_encode_forward");
            case 4: if (strided) hipLaunchKernelGGL((k_grid_forward_pair<T, D, 4, true>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    else hipLaunchKernelGGL((k_grid_forward_pair<T, D, 4>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    return check_launch("grid_encode_forward");
            case 8: if (strided) hipLaunchKernelGGL((k_grid_forward_pair<T, D, 8, true>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    else hipLaunchKernelGGL((k_grid_forward_pair<T, D, 8>), grid_pair, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, gridtype, ac, interp, plan, emb_b, outputs_b);
                    return check_launch("grid_encode_forward");
            default: break;
        }
    }
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_forward<T, D, 1>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        case 2: hipLaunchKernelGGL((k_grid_forward<T, D, 2>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        case 4: hipLaunchKernelGGL((k_grid_forward<T, D, 4>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        case 8: hipLaunchKernelGGL((k_grid_forward<T, D, 8>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, sc, dy_dx, gridtype, ac, interp); break;
        default: set_error("GridEncoding: C must be 1, 2, 4, or 8."); return S3D_ERR_UNSUPPORTED;
    }
    return check_launch("grid_encode_forward");
}

constexpr uint32_t kBinnedMinPoints = 8192;   // below this the direct-atomic kernel wins (fixed cost of the sorted path)
constexpr size_t kBinPassBytes = 4ull << 30;  // record storage per pass; levels are processed in groups that fit

// Workspace layout of the binned path (offsets 256-byte aligned):
//   hdr: |grad| max per level [kMaxLevels] | tot[L][smax] | cursor[L][smax] | keys | vals      (hdr..cursor are memset)
struct BinLayout {
    uint32_t levels_per_pass, chunks, chunk_points, smax;
    size_t tot, cursor, keys, vals, total;
    bool ok;
};
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline BinLayout bin_layout(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level_rows, uint32_t elem) {
    BinLayout o{};
    if (!(C == 1 || C == 2 || C == 4 || C == 8) || D < 2 || D > 5 || !max_level_rows || !L || L > kMaxLevels || !B) return o;
    const uint32_t rec = 4 + elem * C;
    o.smax = bin_slices(max_level_rows, C);  // bin_slices is monotone in rows: no level has more slices
    o.ok = o.smax <= kBinMaxSlices && ((uint64_t)B << D) < (1ull << 31);
    if (!o.ok) return o;
    o.chunk_points = S3D_BIN_CHUNK;  // = bin_chunk_points<T, D, C>()
    o.chunks = div_up<uint32_t>(B, o.chunk_points);
    const size_t per_level = ((size_t)B << D) * rec;
    const size_t lp = kBinPassBytes / per_level;
    o.levels_per_pass = (uint32_t)(lp < 1 ? 1 : (lp > L ? L : lp));
    o.tot = 256;
    o.cursor = o.tot + (size_t)L * o.smax * 4;
    o.keys = align256(o.cursor + (size_t)L * o.smax * 4);
    o.vals = align256(o.keys + (size_t)o.levels_per_pass * ((size_t)B << D) * 4);
    o.total = align256(o.vals + (size_t)o.levels_per_pass * ((size_t)B << D) * elem * C);
    return o;
}

// third-generation layout: hdr[kMaxLevels] | cursor[L][smax][NSUB] | ovn[L][smax] | ovl[L][smax][nchunks] (8 B) |
//   bucket keys [levels_per_pass][smax][NSUB][cap] (2 B) | bucket values (4 B) | spill keys [levels_per_pass][nchunks][P * K] | spill values
struct BinLayout3 {
    uint32_t levels_per_pass, chunks, smax, cap;
    size_t cursor, ovn, ovl, keys, vals, skeys, svals, total;
    bool ok;
};
inline BinLayout3 bin_layout3(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level_rows, uint32_t elem) {
    BinLayout3 o{};
    constexpr uint32_t P = S3D_BIN3_P;
    if (elem * C != 4 || D < 2 || D > 5 || !max_level_rows || !L || L > kMaxLevels || !B) return o;
    o.smax = bin_slices(max_level_rows, C);
    o.ok = o.smax <= kBinMaxSlices && ((uint64_t)B << D) < (1ull << 30) && ((uint64_t)P << D) < 65536 &&
           kBinAccBytes / (8 * C) <= 65536;
    if (!o.ok) return o;
    o.chunks = div_up<uint32_t>(B, P);
    // per sub-bucket: 2x the mean of a uniformly hit level without any merging, split over the XCDs, + slack for tiny batches
    o.cap = (uint32_t)((((uint64_t)B << (D + 1)) / kBinMinSlices / kBin3Sub + 256 + 63) & ~63ull);
    const size_t recs = (size_t)o.smax * kBin3Sub * o.cap, srecs = (size_t)o.chunks * P << D;
    const size_t per_level = (recs + srecs) * 6;
    const size_t lp = kBinPassBytes / per_level;
    o.levels_per_pass = (uint32_t)(lp < 1 ? 1 : (lp > L ? L : lp));
    o.cursor = 256;
    o.ovn = o.cursor + (size_t)L * o.smax * kBin3Sub * kCurStride * 4;
    o.ovl = align256(o.ovn + (size_t)L * o.smax * 4);
    o.keys = align256(o.ovl + (size_t)L * o.smax * o.chunks * 8);
    o.vals = align256(o.keys + (size_t)o.levels_per_pass * recs * 2);
    o.skeys = align256(o.vals + (size_t)o.levels_per_pass * recs * 4);
    o.svals = align256(o.skeys + (size_t)o.levels_per_pass * srecs * 2);
    o.total = align256(o.svals + (size_t)o.levels_per_pass * srecs * 4);
    return o;
}

// CUs of the current device (workgroups of the persistent accumulate), cached per device ordinal
inline uint32_t device_cus() {
    static std::atomic<uint32_t> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    uint32_t n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? (uint32_t)v : 256u;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// control block of the third generation: hdr[kMaxLevels] | ticket (byte 128) | cursor[L][smax][NSUB] (x kCurStride words) | ovn[L][smax]
inline size_t bin3_control_bytes(const BinLayout3& lay, uint32_t L) { return align256(lay.ovn + (size_t)L * lay.smax * 4); }

template <typename T, uint32_t D, uint32_t C, bool FIXED24>
int launch_binned3(const T* grad, const float* inputs, const int32_t* offsets, T* grad_emb, uint32_t B, uint32_t L,
                   const LevelScales& sc, uint32_t gridtype, bool ac, uint32_t interp, unsigned char* ws, const BinLayout3& lay,
                   unsigned char* control, hipStream_t st) {
    constexpr uint32_t P = S3D_BIN3_P;
    constexpr uint32_t K = 1u << D;
    constexpr uint32_t stage_max = 4 * kBinMaxSlices * 4 + P * K * 8;  // counters, run starts, run table (8 B), staged records
    const uint32_t stage = 4 * lay.smax * 4 + P * K * 8;
    static std::atomic<uint64_t> attr_devs{0};
    int dev;
    if (device_needs_setup(attr_devs, &dev)) {
        S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter6<T, D, C, FIXED24, P>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_max));
        S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_accumulate6<T, D, C, FIXED24, P>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinAccBytes));
        device_setup_done(attr_devs, dev);
    }
    // a caller-owned control block is all zero between calls (the accumulate leaves it so); without one the words live at
    // the head of the workspace and are cleared by a launch of their own
    unsigned char* cb = control ? control : ws;
    uint32_t* hdr = reinterpret_cast<uint32_t*>(cb);
    uint32_t* done = reinterpret_cast<uint32_t*>(cb + 128);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(cb + lay.cursor);
    uint32_t* ovn = reinterpret_cast<uint32_t*>(cb + lay.ovn);
    uint2* ovl = reinterpret_cast<uint2*>(ws + lay.ovl);
    uint16_t* keys = reinterpret_cast<uint16_t*>(ws + lay.keys);
    uint32_t* vals = reinterpret_cast<uint32_t*>(ws + lay.vals);
    uint16_t* skeys = reinterpret_cast<uint16_t*>(ws + lay.skeys);
    uint32_t* svals = reinterpret_cast<uint32_t*>(ws + lay.svals);
    if (!control) {
        const uint32_t clear_words = (uint32_t)(bin3_control_bytes(lay, L) / 4);
        hipLaunchKernelGGL(k_zero_words, dim3(div_up<uint32_t>(clear_words, 1024)), dim3(1024), 0, st, hdr, clear_words);
    }
    if constexpr (!FIXED24)
        hipLaunchKernelGGL((k_bin_amax<T, D, C>), dim3(std::min<uint32_t>(div_up<uint32_t>(B, 1024), 64u), L), dim3(1024), 0, st, grad,
                           inputs, B, sc, hdr);
    const uint32_t cus = device_cus();
    for (uint32_t l0 = 0; l0 < L; l0 += lay.levels_per_pass) {
        const uint32_t nl = (L - l0 < lay.levels_per_pass) ? L - l0 : lay.levels_per_pass;
        hipLaunchKernelGGL((k_bin_scatter6<T, D, C, FIXED24, P>), dim3(lay.chunks, nl), dim3(P), stage, st, grad, inputs, offsets, B, l0,
                           sc, hdr, cursor, ovn, ovl, lay.smax, lay.chunks, lay.cap, keys, vals, skeys, svals, gridtype, ac, interp);
        const uint32_t items = lay.smax * nl;
        bool fused = false;
        if constexpr (FIXED24 && sizeof(T) == 2 && C == 2) {
            if (t_adam && nl == L) {  // (one pass covers every level: the skip decision needs all poison words up front)
                static std::atomic<uint64_t> attr_adam{0};
                int dev2;
                if (device_needs_setup(attr_adam, &dev2)) {
                    S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_accumulate6<T, D, C, FIXED24, P, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinAccBytes));
                    device_setup_done(attr_adam, dev2);
                }
                hipLaunchKernelGGL((k_bin_accumulate6<T, D, C, FIXED24, P, true>), dim3(std::min(items, cus * kBinAccPerCu)), dim3(kBinAccThreads),
                                   kBinAccBytes, st, (const uint16_t*)keys, (const uint32_t*)vals, (const uint16_t*)skeys, (const uint32_t*)svals,
                                   offsets, grad_emb, B, l0, nl, hdr, done, cursor, ovn, (const uint2*)ovl, lay.smax, lay.chunks, lay.cap,
                                   t_found_inf, *t_adam);
                fused = true;
                t_adam_applied = true;
            }
        }
        if (!fused)
        hipLaunchKernelGGL((k_bin_accumulate6<T, D, C, FIXED24, P>), dim3(std::min(items, cus * kBinAccPerCu)), dim3(kBinAccThreads), kBinAccBytes, st,
                           (const uint16_t*)keys, (const uint32_t*)vals, (const uint16_t*)skeys, (const uint32_t*)svals, offsets,
                           grad_emb, B, l0, nl, hdr, done, cursor, ovn, (const uint2*)ovl, lay.smax, lay.chunks, lay.cap, t_found_inf, GridAdam{});
    }
    t_reported = true;
    return check_launch("grid_encode_backward");
}

template <typename T, uint32_t D, uint32_t C>
int launch_backward_c(const T* grad, const float* inputs, const int32_t* offsets, uint32_t max_level_rows, T* grad_emb,
                      uint32_t B, uint32_t L, const LevelScales& sc, const T* dy_dx, T* grad_inputs, uint32_t gridtype, bool ac,
                      uint32_t interp, unsigned char* ws, size_t ws_bytes, int force_path, unsigned char* control, size_t control_bytes,
                      hipStream_t st) {
    const BinLayout lay = bin_layout(B, D, C, L, max_level_rows, sizeof(T));
    const bool bin_ok = lay.ok && ws && ws_bytes >= lay.total;
    // 0 auto: binned for large batches; 1 direct atomics; 2 binned
    const bool binned = bin_ok && (force_path >= 2 || (force_path == 0 && B >= kBinnedMinPoints));
    if constexpr (sizeof(T) * C == 4) {  // one 32-bit value word per record (fp16 C = 2, fp32 C = 1)
        const BinLayout3 lay3 = bin_layout3(B, D, C, L, max_level_rows, sizeof(T));
        if (binned && lay3.ok && ws && ws_bytes >= lay3.total) {  // 6-byte records, XCD-private sub-buckets
            int rc;
            unsigned char* cb = (control && control_bytes >= bin3_control_bytes(lay3, L)) ? control : nullptr;
            if (sizeof(T) == 2 && ((uint64_t)B << D) <= (1ull << 23))
                rc = launch_binned3<T, D, C, true>(grad, inputs, offsets, grad_emb, B, L, sc, gridtype, ac, interp, ws, lay3, cb, st);
            else
                rc = launch_binned3<T, D, C, false>(grad, inputs, offsets, grad_emb, B, L, sc, gridtype, ac, interp, ws, lay3, cb, st);
            if (rc != S3D_OK) return rc;
            if (dy_dx && grad_inputs)
                hipLaunchKernelGGL((k_grid_input_backward<T, D, C>), dim3(div_up<uint32_t>(B * D, 256)), dim3(256), 0, st, grad,
                                   dy_dx, grad_inputs, B, L, sc.n_valid);
            return check_launch("grid_encode_backward");
        }
    }
    if (binned) {
        using V = typename FeatVec<T, C>::type;
        constexpr uint32_t P = bin_chunk_points<T, D, C>();
        constexpr uint32_t K = 1u << D;
        constexpr uint32_t stage = 2 * kBinMaxSlices * 4;  // slice counters + reserved bucket offsets
        static std::atomic<uint64_t> attr_devs{0};
        int dev;
        if (device_needs_setup(attr_devs, &dev)) {
            S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter<T, D, C>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage));
            S3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_accumulate<T, D, C>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinAccBytes));
            device_setup_done(attr_devs, dev);
        }
        uint32_t* hdr = reinterpret_cast<uint32_t*>(ws);
        uint32_t* tot = reinterpret_cast<uint32_t*>(ws + lay.tot);
        uint32_t* cursor = reinterpret_cast<uint32_t*>(ws + lay.cursor);
        uint32_t* keys = reinterpret_cast<uint32_t*>(ws + lay.keys);
        V* vals = reinterpret_cast<V*>(ws + lay.vals);
        // (a kernel, not hipMemsetAsync: captured as a hipGraph memset node the clear did not take effect from the second
        //  replay on — stale cursors, out-of-range record writes — on ROCm 7.2)
        const uint32_t clear_words = (uint32_t)((lay.cursor + (size_t)L * lay.smax * 4) / 4);
        hipLaunchKernelGGL(k_zero_words, dim3(div_up<uint32_t>(clear_words, 1024)), dim3(1024), 0, st, hdr, clear_words);
        const uint32_t ppb = div_up<uint32_t>(div_up<uint32_t>(B, kBinCountChunks), kBinCountThreads) * kBinCountThreads;
        hipLaunchKernelGGL((k_bin_count<T, D, C>), dim3(div_up<uint32_t>(B, ppb), L), dim3(kBinCountThreads), 0, st, grad, inputs,
                           offsets, B, ppb, sc, hdr, tot, lay.smax, gridtype, ac, interp);
        for (uint32_t l0 = 0; l0 < L; l0 += lay.levels_per_pass) {
            const uint32_t nl = (L - l0 < lay.levels_per_pass) ? L - l0 : lay.levels_per_pass;
            hipLaunchKernelGGL((k_bin_scatter<T, D, C>), dim3(lay.chunks, nl), dim3(P / kBinQuad), stage, st, grad, inputs, offsets, B, l0, sc,
                               (const uint32_t*)hdr, (const uint32_t*)tot, cursor, lay.smax, keys, vals, gridtype, ac, interp);
            hipLaunchKernelGGL((k_bin_accumulate<T, D, C>), dim3(lay.smax, nl), dim3(kBinAccThreads), kBinAccBytes, st,
                               (const uint32_t*)keys, (const V*)vals, offsets, grad_emb, B, l0, (const uint32_t*)hdr,
                               (const uint32_t*)tot, lay.smax);
        }
    } else {
        hipLaunchKernelGGL((k_grid_backward<T, D, C>), dim3(xcd_grid(B)), dim3(kFwdBlock), 0, st, grad, inputs, offsets,
                           grad_emb, B, L, sc, gridtype, ac, interp);
    }
    if (dy_dx && grad_inputs)
        hipLaunchKernelGGL((k_grid_input_backward<T, D, C>), dim3(div_up<uint32_t>(B * D, 256)), dim3(256), 0, st, grad,
                           dy_dx, grad_inputs, B, L, sc.n_valid);
    return check_launch("grid_encode_backward");
}

template <typename T, uint32_t D>
int launch_backward(const T* grad, const float* inputs, const int32_t* offsets, uint32_t max_level_rows,
                    T* grad_emb, uint32_t B, uint32_t C, uint32_t L, const LevelScales& sc, const T* dy_dx, T* grad_inputs, uint32_t gridtype,
                    bool ac, uint32_t interp, unsigned char* ws, size_t ws_bytes, int force_path, unsigned char* control, size_t control_bytes,
                    hipStream_t st) {
    switch (C) {
        case 1:
            if constexpr (sizeof(T) == 2) {
                set_error("GridEncoding: fp16 tables need an even C (the reference forces fp32 when C is odd, grid.py:42)");
                return S3D_ERR_UNSUPPORTED;
            } else return launch_backward_c<T, D, 1>(grad, inputs, offsets, max_level_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, control, control_bytes, st);
        case 2: return launch_backward_c<T, D, 2>(grad, inputs, offsets, max_level_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, control, control_bytes, st);
        case 4: return launch_backward_c<T, D, 4>(grad, inputs, offsets, max_level_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, control, control_bytes, st);
        case 8: return launch_backward_c<T, D, 8>(grad, inputs, offsets, max_level_rows, grad_emb, B, L, sc, dy_dx, grad_inputs, gridtype, ac, interp, ws, ws_bytes, force_path, control, control_bytes, st);
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

#ifdef S3D_BIN3_PROF
S3D_EXPORT int s3d_debug_prof_read(unsigned long long* dst, size_t bytes, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return S3D_ERR_HIP;
    if (bytes > sizeof(unsigned long long) * 2 * 16384 * 8) bytes = sizeof(unsigned long long) * 2 * 16384 * 8;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(s3d_prof_buf), bytes) != hipSuccess) return S3D_ERR_HIP;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(s3d_prof_buf)) != hipSuccess) return S3D_ERR_HIP;
        if (hipMemset(p, 0, sizeof(unsigned long long) * 2 * 16384 * 8) != hipSuccess) return S3D_ERR_HIP;
    }
    return S3D_OK;
}
#endif

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
                                       float bound, const int32_t* n_valid, const float* live, uint32_t live_stride,
                                       s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && embeddings && offsets && outputs, "grid_encode_forward: null pointer");
    S3D_REQUIRE(!(live && dy_dx), "grid_encode_forward: `live` is an inference-time option (no input Jacobian)");
    S3D_REQUIRE(bound >= 0.0f && !(bound != 0.0f && dy_dx), "grid_encode_forward: bound must be >= 0 (0 = inputs in [0,1]); "
                "the input Jacobian is only produced for pre-normalised inputs");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward: L must be in [1, %u]", kMaxLevels);
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "grid_encode_forward: dtype must be f32 or f16");
    S3D_REQUIRE((uint64_t)B * L * C < (1ull << 32), "grid_encode_forward: B*L*C overflows 32 bits");
    LevelScales sc;
    host_scales(L, S, H, sc, bound, n_valid);
    sc.live = live;
    sc.live_stride = live ? (live_stride ? live_stride : 1u) : 0u;
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

S3D_EXPORT int s3d_grid_encode_forward_pair(const float* inputs, const void* embeddings_a, const void* embeddings_b,
                                            const int32_t* offsets, void* outputs_a, void* outputs_b, uint32_t B, uint32_t D,
                                            uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                            uint32_t interp, int dtype, float bound, const int32_t* n_valid, const float* live,
                                            uint32_t live_stride, s3d_stream_t stream) {
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(inputs && embeddings_a && embeddings_b && offsets && outputs_a && outputs_b, "grid_encode_forward_pair: null pointer");
    S3D_REQUIRE(bound >= 0.0f, "grid_encode_forward_pair: bound must be >= 0 (0 = inputs in [0,1])");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_forward_pair: L must be in [1, %u]", kMaxLevels);
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "grid_encode_forward_pair: dtype must be f32 or f16");
    S3D_REQUIRE((uint64_t)B * L * C < (1ull << 32), "grid_encode_forward_pair: B*L*C overflows 32 bits");
    LevelScales sc;
    host_scales(L, S, H, sc, bound, n_valid);
    sc.live = live;
    sc.live_stride = live ? (live_stride ? live_stride : 1u) : 0u;
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    if (dtype == S3D_F32) {
        const float* e = (const float*)embeddings_a; float* o = (float*)outputs_a;
        const float* e2 = (const float*)embeddings_b; float* o2 = (float*)outputs_b;
        S3D_DISPATCH_D(D, (launch_forward<float, 2>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)),
                       (launch_forward<float, 3>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)),
                       (launch_forward<float, 4>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)),
                       (launch_forward<float, 5>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)))
    } else {
        const __half* e = (const __half*)embeddings_a; __half* o = (__half*)outputs_a;
        const __half* e2 = (const __half*)embeddings_b; __half* o2 = (__half*)outputs_b;
        S3D_DISPATCH_D(D, (launch_forward<__half, 2>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)),
                       (launch_forward<__half, 3>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)),
                       (launch_forward<__half, 4>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)),
                       (launch_forward<__half, 5>(inputs, e, offsets, o, B, C, L, sc, nullptr, gridtype, ac, interp, st, e2, o2)))
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

S3D_EXPORT size_t s3d_grid_encode_backward_workspace_size(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level_rows,
                                                          int dtype) {
    const BinLayout lay = bin_layout(B, D, C, L, max_level_rows, dtype == S3D_F16 ? 2u : 4u);
    const BinLayout3 lay3 = bin_layout3(B, D, C, L, max_level_rows, dtype == S3D_F16 ? 2u : 4u);
    const size_t a = lay.ok ? lay.total : 0, c = lay3.ok ? lay3.total : 0;
    return std::max(a, c);
}

S3D_EXPORT size_t s3d_grid_encode_backward_control_size(uint32_t D, uint32_t C, uint32_t L, uint32_t max_level_rows, int dtype) {
    const BinLayout3 lay3 = bin_layout3(1u << 16, D, C, L, max_level_rows, dtype == S3D_F16 ? 2u : 4u);  // (the block does not depend on B)
    return lay3.ok ? bin3_control_bytes(lay3, L) : 0;
}

S3D_EXPORT int s3d_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                        const int32_t* offsets, void* grad_embeddings,
                                        uint32_t max_level_rows, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                        uint32_t interp, int dtype, void* workspace, size_t workspace_bytes,
                                        float bound, const int32_t* n_valid, int path, float* found_inf,
                                        void* control, size_t control_bytes, s3d_stream_t stream) {
    // path: 0 = auto (binned from 8,192 points), 1 = direct global atomics, 2 = binned (partition + LDS accumulate)
    (void)embeddings;
    S3D_REQUIRE(path >= 0 && path <= 2, "grid_encode_backward: path must be 0 (auto), 1 (atomics) or 2 (binned)");
    S3D_REQUIRE(bound >= 0.0f && !(bound != 0.0f && dy_dx), "grid_encode_backward: bound must be >= 0 and 0 with an input Jacobian");
    if (B == 0) return S3D_OK;
    S3D_REQUIRE(grad && inputs && offsets && grad_embeddings, "grid_encode_backward: null pointer");
    S3D_REQUIRE(L >= 1 && L <= kMaxLevels, "grid_encode_backward: L must be in [1, %u]", kMaxLevels);
    S3D_REQUIRE(dtype == S3D_F32 || dtype == S3D_F16, "grid_encode_backward: dtype must be f32 or f16");
    if (path >= 2) {
        const BinLayout lay = bin_layout(B, D, C, L, max_level_rows, dtype == S3D_F16 ? 2u : 4u);
        S3D_REQUIRE(lay.ok && workspace && workspace_bytes >= lay.total,
                    "grid_encode_backward: the binned path needs max_level_rows and a workspace of "
                    "s3d_grid_encode_backward_workspace_size() bytes");
    }
    LevelScales sc;
    host_scales(L, S, H, sc, bound, n_valid);
    hipStream_t st = as_stream(stream);
    const bool ac = align_corners != 0;
    unsigned char* ws = (unsigned char*)workspace;
    unsigned char* cb = (unsigned char*)control;
    const int fp = path;
    t_found_inf = found_inf;
    t_reported = false;
    const auto run = [&]() -> int {
    if (dtype == S3D_F32) {
        const float* g = (const float*)grad; float* ge = (float*)grad_embeddings;
        const float* j = (const float*)dy_dx; float* gi = (float*)grad_inputs;
        S3D_DISPATCH_D(D, (launch_backward<float, 2>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)),
                       (launch_backward<float, 3>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)),
                       (launch_backward<float, 4>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)),
                       (launch_backward<float, 5>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)))
    } else {
        const __half* g = (const __half*)grad; __half* ge = (__half*)grad_embeddings;
        const __half* j = (const __half*)dy_dx; __half* gi = (__half*)grad_inputs;
        S3D_DISPATCH_D(D, (launch_backward<__half, 2>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)),
                       (launch_backward<__half, 3>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)),
                       (launch_backward<__half, 4>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)),
                       (launch_backward<__half, 5>(g, inputs, offsets, max_level_rows, ge, B, C, L, sc, j, gi, gridtype, ac, interp, ws, workspace_bytes, fp, cb, control_bytes, st)))
    }
    };
    const int rc = run();
    t_found_inf = nullptr;
    if (rc != S3D_OK || !found_inf || t_reported) return rc;
    if (dtype == S3D_F32)
        hipLaunchKernelGGL(k_table_nonfinite<float>, dim3(kMaxStreamBlocks), dim3(256), 0, st, (const float*)grad_embeddings, offsets, L, C, found_inf);
    else
        hipLaunchKernelGGL(k_table_nonfinite<__half>, dim3(kMaxStreamBlocks), dim3(256), 0, st, (const __half*)grad_embeddings, offsets, L, C, found_inf);
    return check_launch("grid_encode_backward (gradient check)");
}

S3D_EXPORT int s3d_grid_encode_backward_adam(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                             void* grad_embeddings, uint32_t max_level_rows, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                             float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                             void* workspace, size_t workspace_bytes, float bound, const int32_t* n_valid, float* found_inf,
                                             void* control, size_t control_bytes, const s3d_grid_adam* adam, int* applied,
                                             s3d_stream_t stream) {
    S3D_REQUIRE(adam && applied, "grid_encode_backward_adam: null pointer");
    S3D_REQUIRE(adam->param && adam->exp_avg && adam->exp_avg_sq && adam->step, "grid_encode_backward_adam: null optimizer state");
    *applied = 0;
    GridAdam ga;
    ga.p = adam->param; ga.m = adam->exp_avg; ga.v = adam->exp_avg_sq; ga.ph = reinterpret_cast<__half*>(adam->param_half);
    ga.lr = adam->lr; ga.beta1 = adam->beta1; ga.beta2 = adam->beta2; ga.eps = adam->eps;
    ga.step = adam->step; ga.grad_scale = adam->grad_scale; ga.lr_scale = adam->lr_scale;
    t_adam = &ga;
    t_adam_applied = false;
    const int rc = s3d_grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, max_level_rows, B, D, C, L, S, H, nullptr,
                                            nullptr, gridtype, align_corners, interp, dtype, workspace, workspace_bytes, bound, n_valid, 0,
                                            found_inf, control, control_bytes, stream);
    t_adam = nullptr;
    *applied = t_adam_applied ? 1 : 0;
    t_adam_applied = false;
    return rc;
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
