// raymarching.hip — occupancy-grid ray marching, morton codes, bit packing and volume
// compositing for gfx950.  Replaces raymarching/src/raymarching.cu of the reference behind the
// C ABI of include/seal3d_hip.h (per-function citations there).
//
// MI355X notes
//  * One ray per lane, ONE WAVE PER WORKGROUP for the marching kernels: a 4,096-ray training
//    batch is only 64 waves, so the kernels are latency-bound on the per-step bitfield probe;
//    64-thread workgroups spread those waves over 64 CUs instead of 16.
//  * Ray compaction: spans are reserved with a wave64 prefix sum (no per-ray atomics).  The
//    count kernel leaves (local offset, count) per ray and one total per wave; the write
//    kernel turns wave totals into the wave's base with a strided wave reduction.  The packing
//    is therefore the RAY-ORDERED one — deterministic, bit-identical to the oracle.
//  * Integer results (cell coordinates, morton rows, sample counts, span offsets) follow the
//    oracle expression by expression: explicit fmaf, -ffp-contract=off, IEEE division.
#include "s3d_common.hpp"

namespace s3d {
namespace {

constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kRPi = 0.3183098861837907f;

// raymarching.cu:56-63 as written: multiplies by 0x00010001, 0x101, 0x11, 5 with uint32 wrap-around — what the exported
// morton3D kernel computes for ANY int32 input (tests/golden/int_kernels.npz holds inputs up to 2^32 - 1).
__device__ __forceinline__ uint32_t expand_bits_any(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
// The marching kernels' form, for v < 2^16 (cell coordinates are clamped to [0, H - 1], H <= 1024): the shifted copy never
// overlaps the original there — in the first step because v has no bit above 15, afterwards because of the masks — so
// `v * (1 + 2^k)` == `v | v << k` bit for bit, and a 32-bit integer multiply is a quarter-rate instruction here while
// shift-or is one full-rate v_lshl_or_b32: 12 multiplies per occupancy probe gone.  (v >= 2^16: the product carries where
// the `or` does not — the exported kernel therefore uses expand_bits_any.)
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v | (v << 16)) & 0xFF0000FFu;
    v = (v | (v << 8)) & 0x0F00F00Fu;
    v = (v | (v << 4)) & 0xC30C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t morton3d_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// ---------------------------------------------------------------- small utilities
// counter-based per-ray jitter for a graph-replayed step: u01(key, step, ray) with the step number read from device memory
// (torch.rand inside a captured graph costs its own kernel plus two seed/offset fills before every replay)
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    const uint32_t w = ((v >> ((v >> 28u) + 4u)) ^ v) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ float ray_noise(uint32_t key, uint32_t step, uint32_t n) {
    return (float)(pcg_hash(pcg_hash(key ^ (step * 0x9E3779B9u)) + n) >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

// Rows of a padded sample batch that no ray fills but the consumers still process (seal3d_hip.h: `n_valid` rounds the sample
// count up to 128 rows; a ray that does not fit the budget M leaves [offset, M) empty).  The training kernels write them as
// zeros themselves, so the caller's buffers need no zero fill: `lane`/`nlanes` = the lanes sharing the work.
__device__ __forceinline__ uint32_t pad_end(uint32_t total, uint32_t M) {
    const uint32_t e = (total + 127u) & ~127u;
    return e < M ? e : M;
}
__device__ __forceinline__ void zero_sample_rows(float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                                                 uint32_t lo, uint32_t hi, uint32_t lane, uint32_t nlanes) {
    for (size_t o = (size_t)lo + lane; o < hi; o += nlanes) {
        xyzs[o * 3] = 0.0f; xyzs[o * 3 + 1] = 0.0f; xyzs[o * 3 + 2] = 0.0f;
        dirs[o * 3] = 0.0f; dirs[o * 3 + 1] = 0.0f; dirs[o * 3 + 2] = 0.0f;
        deltas[o * 2] = 0.0f; deltas[o * 2 + 1] = 0.0f;
    }
}
__device__ __forceinline__ void zero_grad_rows(float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs, uint32_t lo,
                                               uint32_t hi, uint32_t lane, uint32_t nlanes) {
    for (size_t o = (size_t)lo + lane; o < hi; o += nlanes) {
        grad_sigmas[o] = 0.0f;
        grad_rgbs[o * 3] = 0.0f; grad_rgbs[o * 3 + 1] = 0.0f; grad_rgbs[o * 3 + 2] = 0.0f;
    }
}

// raymarching.cu:near_far_from_aabb — slab test against the box, near clamped to min_near; a miss gives FLT_MAX twice
__device__ __forceinline__ void near_far_of(const float* __restrict__ rays_o, const float* __restrict__ rays_d, uint32_t n,
                                            const float (&a)[6], float min_near, float& near_out, float& far_out) {
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float rdx = 1 / rays_d[n * 3], rdy = 1 / rays_d[n * 3 + 1], rdz = 1 / rays_d[n * 3 + 2];
    float near = (a[0] - ox) * rdx, far = (a[3] - ox) * rdx;
    if (near > far) { float t = near; near = far; far = t; }
    float near_y = (a[1] - oy) * rdy, far_y = (a[4] - oy) * rdy;
    if (near_y > far_y) { float t = near_y; near_y = far_y; far_y = t; }
    bool miss = (near > far_y || near_y > far);
    if (!miss) {
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (a[2] - oz) * rdz, far_z = (a[5] - oz) * rdz;
        if (near_z > far_z) { float t = near_z; near_z = far_z; far_z = t; }
        miss = (near > far_z || near_z > far);
        if (!miss) {
            if (near_z > near) near = near_z;
            if (far_z < far) far = far_z;
            if (near < min_near) near = min_near;
        }
    }
    near_out = miss ? 3.402823466e+38f : near;
    far_out = miss ? 3.402823466e+38f : far;
}

__global__ void k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                           const float* __restrict__ aabb, uint32_t N, float min_near,
                           float* __restrict__ nears, float* __restrict__ fars, float* __restrict__ noises,
                           const int32_t* __restrict__ noise_step, uint32_t noise_key) {
    const float a[6] = {aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5]};
    const uint32_t step = (noises && noise_step) ? (uint32_t)*noise_step : 0u;
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        if (noises) noises[n] = ray_noise(noise_key, step, n);
        float near, far;
        near_far_of(rays_o, rays_d, n, a, min_near, near, far);
        nears[n] = near;
        fars[n] = far;
    }
}

__global__ void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                               float radius, uint32_t N, float* __restrict__ coords) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const float B = __builtin_fmaf(oz, dz, __builtin_fmaf(oy, dy, ox * dx));
        const float Cc = __builtin_fmaf(-radius, radius, __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox)));
        const float t = (-B + sqrtf(__builtin_fmaf(B, B, -(A * Cc)))) / A;
        const float x = __builtin_fmaf(t, dx, ox), y = __builtin_fmaf(t, dy, oy), z = __builtin_fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(__builtin_fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = __builtin_fmaf(2 * theta, kRPi, -1.0f);
        coords[n * 2 + 1] = phi * kRPi;
    }
}

__global__ void k_morton3d(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x)
        indices[n] = (int32_t)(expand_bits_any((uint32_t)coords[n * 3]) | (expand_bits_any((uint32_t)coords[n * 3 + 1]) << 1) |
                               (expand_bits_any((uint32_t)coords[n * 3 + 2]) << 2));
}

__global__ void k_morton3d_invert(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const int32_t ind = indices[n];
        coords[n * 3] = (int32_t)morton3d_invert((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)morton3d_invert((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)morton3d_invert((uint32_t)(ind >> 2));
    }
}

// one lane = one output byte = 8 cells = two float4 loads (32 B/lane, 2 KiB contiguous per wave)
__global__ void k_packbits(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield) {
    const float4* g4 = reinterpret_cast<const float4*>(grid);
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float4 a = g4[n * 2], b = g4[n * 2 + 1];
        uint32_t bits = 0;
        bits |= (a.x > thresh) ? 1u : 0u;   bits |= (a.y > thresh) ? 2u : 0u;
        bits |= (a.z > thresh) ? 4u : 0u;   bits |= (a.w > thresh) ? 8u : 0u;
        bits |= (b.x > thresh) ? 16u : 0u;  bits |= (b.y > thresh) ? 32u : 0u;
        bits |= (b.z > thresh) ? 64u : 0u;  bits |= (b.w > thresh) ? 128u : 0u;
        bitfield[n] = (uint8_t)bits;
    }
}

// ---------------------------------------------------------------- occupancy sweep (density-grid maintenance)
// The steady-state update of the reference (nerf/renderer.py:497-538) as three launches around the density query instead of
// ~40 elementwise torch kernels: cells + jittered positions out of two sorted uniform streams, then scatter / EMA-max / mean.
//   draw:   i <  N: cell = floor(u_uniform[i] * H^3)                         (uniform cells; morton index = cell id)
//           i >= N: cell = the floor(u_occupied[i-N] * #occupied)-th cell with density > 0   (binary search in the prefix counts)
//           xyz = (2 c / (H-1) - 1) * (bound - half_cell) + (2 r - 1) * half_cell,  r = counter-based u01(key, step, 3 i + d)
__global__ void __launch_bounds__(256) k_sweep_draw(const double* __restrict__ u_uniform, const double* __restrict__ u_occupied,
                                                    const int32_t* __restrict__ occ_csum, uint32_t N, uint32_t H, float bound,
                                                    float half_cell, uint32_t noise_key, const int32_t* __restrict__ noise_step,
                                                    int32_t* __restrict__ cells, float* __restrict__ xyzs) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * N) return;
    const uint32_t H3 = H * H * H;
    uint32_t cell;
    if (i < N) {
        const double v = u_uniform[i] * (double)H3;
        cell = v >= (double)(H3 - 1) ? H3 - 1 : (uint32_t)v;
    } else {
        const int32_t total = occ_csum[H3 - 1];
        const int32_t pick = (int32_t)(u_occupied[i - N] * (double)total);
        // first cell whose inclusive count exceeds `pick` (torch.searchsorted(csum, pick, right=True)), clamped
        uint32_t lo = 0, hi = H3;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (occ_csum[mid] > pick) hi = mid; else lo = mid + 1;
        }
        cell = lo < H3 ? lo : H3 - 1;
    }
    cells[i] = (int32_t)cell;
    const uint32_t step = noise_step ? (uint32_t)*noise_step : 0u;
    const float inv = 2.0f / (float)(H - 1), span = bound - half_cell;
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        const float c = (float)morton3d_invert(cell >> d);
        const float r = ray_noise(noise_key, step, 3u * i + d);
        xyzs[(size_t)i * 3 + d] = __builtin_fmaf(__builtin_fmaf(2.0f, r, -1.0f), half_cell, (c * inv - 1.0f) * span);
    }
}

__global__ void __launch_bounds__(256) k_sweep_fill(float* __restrict__ tmp, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) tmp[i] = -1.0f;
}
// tmp[cell] = max over the samples of the cell (the reference's indexed assignment keeps an arbitrary one of them; the max is one
// of its outcomes and does not depend on the order).  Densities are >= 0 or NaN: as int32 patterns non-negative floats order
// like the floats and lie above -1.0f; a NaN (0x7fc00000) wins, i.e. poisons the cell exactly like in the reference.
template <typename T>
__global__ void __launch_bounds__(256) k_sweep_scatter(const int32_t* __restrict__ cells, const T* __restrict__ sigma, uint32_t n,
                                                       float scale, float* __restrict__ tmp) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = (float)sigma[i] * scale;
    atomicMax(reinterpret_cast<int32_t*>(tmp) + cells[i], __float_as_int(v));
}
// grid = max(grid * decay, tmp) where both are >= 0 (renderer.py:529-531); per-block sums of clamp(grid, 0) for the mean
__global__ void __launch_bounds__(256) k_sweep_update(float* __restrict__ grid, const float* __restrict__ tmp, uint32_t n, float decay,
                                                      float* __restrict__ partial) {
    __shared__ float wsum[4];
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float g = grid[i];
        const float t = tmp[i];
        if (g >= 0.0f && t >= 0.0f) { g = fmaxf(g * decay, t); grid[i] = g; }
        acc += g > 0.0f ? g : 0.0f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ void __launch_bounds__(256) k_sweep_mean(const float* __restrict__ partial, uint32_t nblocks, float inv_n, float* __restrict__ mean,
                                                    int32_t* __restrict__ step_counter) {
    __shared__ float wsum[4];
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        *mean = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * inv_n;
        if (step_counter) *step_counter += 1;
    }
}

// ---------------------------------------------------------------- DDA core
struct Ray {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};
struct MarchParams {
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Cf, Hf, Hm1, halfH;
    const uint8_t* grid;
};

__device__ __forceinline__ MarchParams make_params(float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                                   uint32_t H, const uint8_t* grid) {
    MarchParams p;
    p.bound = bound; p.dt_gamma = dt_gamma;
    p.dt_min = 2 * kSqrt3 / (float)max_steps;
    p.dt_max = 2 * kSqrt3 * (float)(1 << (C - 1)) / (float)H;
    p.rH = 1 / (float)H;
    p.H3 = (float)(H * H * H);
    p.Cf = (float)C; p.Hf = (float)H; p.Hm1 = (float)(H - 1); p.halfH = 0.5f * (float)H;
    p.grid = grid;
    return p;
}

__device__ __forceinline__ Ray load_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, uint32_t n) {
    Ray r;
    r.ox = rays_o[n * 3]; r.oy = rays_o[n * 3 + 1]; r.oz = rays_o[n * 3 + 2];
    r.dx = rays_d[n * 3]; r.dy = rays_d[n * 3 + 1]; r.dz = rays_d[n * 3 + 2];
    r.rdx = 1 / r.dx; r.rdy = 1 / r.dy; r.rdz = 1 / r.dz;
    return r;
}

// Cascade of a position / of a step (raymarching.cu:42-54): the frexp exponent clamped to [0, C - 1].
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float Cf) {
    int e;
    (void)frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e);
    return (int)fminf(Cf - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float Hf, float Cf) {
    int e;
    (void)frexpf((dt * Hf) * 0.5f, &e);  // (`dt * H * 0.5`: the float product halved in double — halving is exact in float too)
    return (int)fminf(Cf - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_level(float x, float y, float z, float dt, const MarchParams& p) {
    const int la = mip_from_pos(x, y, z, p.Cf), lb = mip_from_dt(dt, p.Hf, p.Cf);
    return la > lb ? la : lb;
}

// test hook: the marcher's cascade selection on arrays (pinned to the reference text by tests/golden/int_kernels.npz)
__global__ void k_mip_levels(const float* __restrict__ xyz, const float* __restrict__ dt, uint32_t N, float Hf, float Cf,
                             int32_t* __restrict__ mip_pos, int32_t* __restrict__ mip_dt) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        if (mip_pos) mip_pos[n] = mip_from_pos(xyz[n * 3], xyz[n * 3 + 1], xyz[n * 3 + 2], Cf);
        if (mip_dt) mip_dt[n] = mip_from_dt(dt[n], Hf, Cf);
    }
}

// Cell of the grid at parameter t (raymarching.cu:358-372): position, step, cascade, cell coordinates; returns the bit index.
struct ProbeCell {
    float x, y, z, dt, mip_bound;
    int nx, ny, nz;
};
__device__ __forceinline__ uint32_t probe_cell(const Ray& r, const MarchParams& p, float t, ProbeCell& c) {
    c.x = clampf(__builtin_fmaf(t, r.dx, r.ox), -p.bound, p.bound);
    c.y = clampf(__builtin_fmaf(t, r.dy, r.oy), -p.bound, p.bound);
    c.z = clampf(__builtin_fmaf(t, r.dz, r.oz), -p.bound, p.bound);
    c.dt = clampf(t * p.dt_gamma, p.dt_min, p.dt_max);
    const int level = mip_level(c.x, c.y, c.z, c.dt, p);
    c.mip_bound = fminf(ldexpf(1.0f, level), p.bound);
    const float mip_rbound = 1 / c.mip_bound;
    // raymarching.cu:366-368 `0.5 * (x * mip_rbound + 1) * H` is a DOUBLE product of the float v = fma(x, mip_rbound, 1) (24
    // significant bits) with the integer H <= 1,024: v * H / 2 has at most 35 bits, so the double holds the exact real number
    // and the conversion back to float rounds it once — which is what ONE fp32 multiply of v by the exactly representable
    // constant H / 2 does as well (IEEE: the correctly rounded exact product).  Bit-identical cells, no fp64 on the hot loop
    // (quarter-rate on CDNA; the wave-per-ray count kernel spent a tenth of its 2,797 vector instructions here).
    c.nx = (int)clampf(__builtin_fmaf(c.x, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
    c.ny = (int)clampf(__builtin_fmaf(c.y, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
    c.nz = (int)clampf(__builtin_fmaf(c.z, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
    return (uint32_t)((float)level * p.H3 + (float)morton3d((uint32_t)c.nx, (uint32_t)c.ny, (uint32_t)c.nz));
}
// parameter at which the ray leaves the (empty) cell of `c` (raymarching.cu:388-394)
__device__ __forceinline__ float probe_exit(const Ray& r, const MarchParams& p, float t, const ProbeCell& c) {
    const float sx = copysignf(1.0f, r.dx), sy = copysignf(1.0f, r.dy), sz = copysignf(1.0f, r.dz);
    const float tx = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.5f, sx, (float)c.nx + 0.5f) * p.rH, 2.0f, -1.0f), c.mip_bound, -c.x) * r.rdx;
    const float ty = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.5f, sy, (float)c.ny + 0.5f) * p.rH, 2.0f, -1.0f), c.mip_bound, -c.y) * r.rdy;
    const float tz = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.5f, sz, (float)c.nz + 0.5f) * p.rH, 2.0f, -1.0f), c.mip_bound, -c.z) * r.rdz;
    return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
}
// Probe the grid at parameter t.  Returns occupancy; x,y,z,dt always valid; t_skip valid when !occ.
__device__ __forceinline__ bool probe(const Ray& r, const MarchParams& p, float t, float& x, float& y, float& z,
                                      float& dt, float& t_skip) {
    ProbeCell c;
    const uint32_t index = probe_cell(r, p, t, c);
    x = c.x; y = c.y; z = c.z; dt = c.dt;
    const bool occ = (p.grid[index >> 3] & (1u << (index & 7u))) != 0;
    if (!occ) t_skip = probe_exit(r, p, t, c);
    return occ;
}

__device__ __forceinline__ float skip_to(const MarchParams& p, float t, float tt) {
    do { t += clampf(t * p.dt_gamma, p.dt_min, p.dt_max); } while (t < tt);
    return t;
}

// ---------------------------------------------------------------- march_rays_train
// workspace: u32 base | u32 pad[3] | u32 wave_total[nw]
constexpr uint32_t kWsHeader = 4;

// pass 1: count + wave-local exclusive scan.  grid = ceil(N/64) workgroups of one wave.
__global__ void __launch_bounds__(64) k_march_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                    const float* __restrict__ nears, const float* __restrict__ fars,
                                                    const float* __restrict__ noises, int32_t* __restrict__ rays,
                                                    const int32_t* __restrict__ counter, uint32_t* __restrict__ ws) {
    const uint32_t n = blockIdx.x * 64 + threadIdx.x;
    uint32_t num_steps = 0;
    if (n < N) {
        const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, grid);
        const Ray r = load_ray(rays_o, rays_d, n);
        const float far = fars[n];
        float t = nears[n];
        t = __builtin_fmaf(clampf(t * dt_gamma, p.dt_min, p.dt_max), noises[n], t);
        while (t < far && num_steps < max_steps) {
            float x, y, z, dt, tt;
            if (probe(r, p, t, x, y, z, dt, tt)) { num_steps++; t += dt; }
            else t = skip_to(p, t, tt);
        }
    }
    const uint32_t incl = wave_incl_scan(num_steps);
    if (n < N) {
        rays[n * 3] = (int32_t)n;
        rays[n * 3 + 1] = (int32_t)(incl - num_steps);  // wave-local offset, rebased by pass 2
        rays[n * 3 + 2] = (int32_t)num_steps;
    }
    if (threadIdx.x == 63) ws[kWsHeader + blockIdx.x] = incl;
    if (blockIdx.x == 0 && threadIdx.x == 0) ws[0] = (uint32_t)counter[0];
}

// pass 2: rebase spans, write samples.
__global__ void __launch_bounds__(64) k_march_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                    const float* __restrict__ nears, const float* __restrict__ fars,
                                                    const float* __restrict__ noises, float* __restrict__ xyzs,
                                                    float* __restrict__ dirs, float* __restrict__ deltas,
                                                    int32_t* __restrict__ rays, int32_t* __restrict__ counter,
                                                    const uint32_t* __restrict__ ws) {
    // wave base = counter value at entry + totals of all earlier waves (strided wave reduction)
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 64) part += ws[kWsHeader + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    const uint32_t base = ws[0] + part;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        counter[0] = (int32_t)(base + ws[kWsHeader + blockIdx.x]);
        counter[1] = counter[1] + (int32_t)N;
    }
    const uint32_t n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
    const uint32_t off = base + (uint32_t)rays[n * 3 + 1];
    rays[n * 3 + 1] = (int32_t)off;
    if (n == N - 1 && off + num_steps < M) zero_sample_rows(xyzs, dirs, deltas, off + num_steps, pad_end(off + num_steps, M), 0, 1);
    if (num_steps != 0 && off < M && off + num_steps > M) zero_sample_rows(xyzs, dirs, deltas, off, M, 0, 1);
    if (num_steps == 0 || off + num_steps > M) return;

    const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, grid);
    const Ray r = load_ray(rays_o, rays_d, n);
    const float far = fars[n];
    float t = nears[n];
    t = __builtin_fmaf(clampf(t * dt_gamma, p.dt_min, p.dt_max), noises[n], t);
    float last_t = t;
    float* px = xyzs + (size_t)off * 3;
    float* pd = dirs + (size_t)off * 3;
    float* pl = deltas + (size_t)off * 2;
    uint32_t step = 0;
    while (t < far && step < num_steps) {
        float x, y, z, dt, tt;
        if (probe(r, p, t, x, y, z, dt, tt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            t += dt;
            pl[0] = dt; pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2; step++;
        } else t = skip_to(p, t, tt);
    }
}


// ---------------------------------------------------------------- march_rays_train, one WAVE per ray
// Measured: the lane-per-ray kernels above take ~250 us per pass for a 4,096-ray batch — 64 waves on a 256-CU
// chip, each serialised on ~200 dependent bitfield probes.  Observation that unlocks parallelism without changing
// a single bit of the result: the sequence of ray parameters t_{k+1} = t_k + clamp(t_k*dt_gamma, dt_min, dt_max)
// does NOT depend on occupancy (the reference advances t by the same increment whether it samples or skips,
// raymarching.cu:386,397); occupancy only decides which t_k are PROBED.  So per ray (one wave):
//   A. one lane generates the t sequence of a 1,024-entry window into LDS (sequential fp32 adds, exact; eight per
//      loop trip, and with dt_gamma == 0 the increment is the constant dt_min — one dependent add per step);
//   B. all 64 lanes probe the occupancy of every t_k in parallel and, for empty probes, find the index the
//      reference's skip loop would land on (first t_m >= t_skip, m > k): next[k] = k+1 (occupied) or m (empty);
//   C. the reference's walk "occupied -> emit, go to k+1; empty -> jump" visits exactly the nodes of the list
//      start -> next[start] -> ...; they are marked by pointer doubling (round i marks next^(2^i) of every marked node,
//      then squares the jump table: ceil(log2(window)) rounds of 64-wide LDS gathers instead of a ~200-step dependent
//      chain on one lane), and the occupied marked nodes are emitted in order with a ballot prefix.
// Emitted t values go to a scratch row; after the ray-ordered prefix sum a second kernel expands them into
// xyz / dirs / deltas with one lane per sample (coalesced).  ~5x more probes than the serial walk, but 64-wide.
constexpr uint32_t kWin = 1024;
constexpr uint16_t kOcc = 0xFFFF;
// workspace (wave path): u32 base | u32 pad[3] | u32 counts[N] | float tsamples[N * max_steps]

template <bool CONST_DT, bool FAST>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_march_count_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                         float* __restrict__ nears, float* __restrict__ fars,
                                                         float* __restrict__ noises, int32_t* __restrict__ rays,
                                                         const int32_t* __restrict__ counter, uint32_t* __restrict__ ws,
                                                         const float* __restrict__ aabb, float min_near,
                                                         const int32_t* __restrict__ noise_step, uint32_t noise_key) {
    __shared__ float T[kWin + 8];
    // (occupancy of the probes lives in a per-lane bit mask — bit i = entry lane + 64 i, like `mk` — not in a third LDS table:
    //  8.5 KB instead of 10.5 KB per single-wave workgroup is 16 instead of 15 resident per CU, i.e. all 4,096 rays of a
    //  training batch in ONE round on 256 CUs instead of a 3,840 + 256 split)
    __shared__ uint16_t J[2][kWin + 2];     // jump tables of the pointer-doubling rounds
    __shared__ uint32_t mk[64];             // visited bits: entry k = lane + 64 i is bit i of mk[lane] (kWin / 64 + 1 <= 32 bits)
    __shared__ uint32_t s_wn;
    __shared__ uint32_t lut[FAST ? 128 : 1];  // FAST: spread coordinates of the morton code (H <= 128)
    const uint32_t n = blockIdx.x, lane = threadIdx.x;
    if constexpr (FAST) {
        lut[lane] = expand_bits(lane);
        lut[lane + 64] = expand_bits(lane + 64);
        // (made visible by the window loop's first barrier, in front of the probes)
    }
    const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, grid);
    const Ray r = load_ray(rays_o, rays_d, n);
    float far, t_carry, noise;
    if (aabb) {  // near / far / jitter of the ray made here (k_near_far's arithmetic) and left for the write pass and the caller
        const float a[6] = {aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5]};
        near_far_of(rays_o, rays_d, n, a, min_near, t_carry, far);
        noise = noises ? (noise_step ? ray_noise(noise_key, (uint32_t)*noise_step, n) : noises[n]) : 0.0f;
        if (lane == 0) {
            nears[n] = t_carry;
            fars[n] = far;
            if (noises && noise_step) noises[n] = noise;
        }
    } else {
        far = fars[n];
        t_carry = nears[n];
        noise = noises[n];
    }
    const float dt0 = clampf(0.0f, p.dt_min, p.dt_max);  // the step when dt_gamma == 0 (dt_max if max_steps is tiny)
    auto dtf = [&](float t) { return CONST_DT ? dt0 : clampf(t * dt_gamma, p.dt_min, p.dt_max); };
    t_carry = __builtin_fmaf(clampf(t_carry * dt_gamma, p.dt_min, p.dt_max), noise, t_carry);
    float* tout = reinterpret_cast<float*>(ws + kWsHeader + N) + (size_t)n * max_steps;

    uint32_t num_steps = 0;       // wave-uniform
    float pending_tt = -INFINITY;  // wave-uniform: skip target carried over from the previous window
#ifdef S3D_MARCH_PROFILE
    long long tp[5] = {0, 0, 0, 0, 0};
    long long c0 = wall_clock64();
#define S3D_TICK(i) { const long long c1 = wall_clock64(); tp[i] += c1 - c0; c0 = c1; }
#else
#define S3D_TICK(i)
#endif
    for (;;) {
        // ---- A: t sequence of this window (T[wn] is the first value past the window or past `far`)
        if constexpr (CONST_DT) {
            // t_{k+1} = fl(t_k + dt0) with a CONSTANT dt0.  Inside one binade every t_k is a multiple of the binade's ulp,
            // so from the second step on the rounded increment is the same number of ulps every time (the only
            // position-dependent case, a tie, resolves to "even" and then stays even): t_k advances by a constant integer
            // step in its IEEE bit pattern.  Two real fp32 adds establish that step; the rest of the binade is filled
            // 64-wide from the bit pattern — the same values the sequential loop produces, without its ~30 ns per step.
            // All lanes run the (uniform) control flow; binade crossings and anything unusual take single real steps.
            float t = t_carry;
            uint32_t k = 0;
            const uint32_t bfar = __float_as_uint(far);
            while (k < kWin && t < far) {
                const float t1 = t + dt0, t2 = t1 + dt0;
                const uint32_t b0 = __float_as_uint(t), b1 = __float_as_uint(t1), b2 = __float_as_uint(t2);
                const uint32_t inc = b2 - b1;
                if (!(t > 0.0f) || (b0 >> 23) != (b2 >> 23) || (b0 >> 23) == 0 || inc == 0 || !(far > 0.0f) || bfar >= 0x7f800000u) {
                    if (lane == 0) T[k] = t;  // one real step
                    k++;
                    t = t1;
                    continue;
                }
                const uint32_t hi = ((b1 >> 23) + 1) << 23;             // first pattern of the next binade
                const uint32_t cnt = (hi - b1) / inc + 1;                // v_j = pattern b1 + j*inc, j < cnt, all <= hi
                const uint32_t nfill = min(cnt, kWin - k);               // positions k+1 .. k+nfill (<= kWin)
                const uint32_t n_lt = bfar <= b1 ? 0u : min(cnt, (bfar - b1 + inc - 1) / inc);  // how many v_j < far
                if (lane == 0) T[k] = t;
                for (uint32_t j = lane; j < nfill; j += 64) T[k + 1 + j] = __uint_as_float(b1 + j * inc);
                if (n_lt < nfill) {  // the ray ends inside this run: T[wn] = v_{n_lt} >= far is already in place
                    k = k + 1 + n_lt;
                    t = __uint_as_float(b1 + n_lt * inc);
                    break;
                }
                // all filled values are below far: the last one starts the next run (or is the carry of a full window)
                k += nfill;
                t = __uint_as_float(b1 + (nfill - 1) * inc);
            }
            if (lane == 0) { T[k] = t; s_wn = k; }
        } else if (lane == 0) {
            float t = t_carry;
            uint32_t k = 0;
            while (k < kWin && t < far) {
                float v[8];
                v[0] = t;
#pragma unroll
                for (uint32_t i = 1; i < 8; i++) v[i] = v[i - 1] + dtf(v[i - 1]);
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) T[k + i] = v[i];
                uint32_t c = 1;  // v is increasing: the entries below `far` are a prefix
#pragma unroll
                for (uint32_t i = 1; i < 8; i++) c += (v[i] < far) ? 1u : 0u;
                if (c < 8) {
                    float tn = v[1];
#pragma unroll
                    for (uint32_t i = 2; i < 8; i++) tn = (c == i) ? v[i] : tn;
                    k += c;
                    t = tn;
                    break;
                }
                k += 8;
                t = v[7] + dtf(v[7]);
            }
            T[k] = t;
            s_wn = k;
        }
        __syncthreads();
        S3D_TICK(0)
        const uint32_t wn = s_wn;
        // ---- B: probe every t_k
        uint32_t occm = 0;  // bit i: entry lane + 64 i is occupied
        uint32_t headm = 0;         // FAST: bit i: entry lane + 64 i is the first of its voxel crossing (its cell differs from entry k - 1's)
        // first m in (k, wn] with T[m] >= tt (m == wn: leaves the window (carry) or the ray, T[wn] >= far): the index starts
        // from the arithmetic estimate (the sequence advances by nearly constant steps) and is walked to the exact entry
        auto skip_index = [&](uint32_t k, float tk, float tt) {
            const float t1 = T[k + 1];
            const float est = (tt - t1) / fmaxf(t1 - tk, 1e-30f);
            uint32_t m = k + 1 + (est > 0.0f ? (est < (float)kWin ? (uint32_t)est : kWin) : 0u);
            m = m < wn ? m : wn;
            while (m > k + 1 && !(T[m - 1] < tt)) m--;
            while (m < wn && T[m] < tt) m++;
            return (uint16_t)m;
        };
        if constexpr (FAST) {
            // one cascade, H <= 128 (every BASELINE configuration): no cascade arithmetic, morton codes from an LDS table of spread
            // coordinates, and TWO passes — the bitfield bytes of all of the lane's probes are requested before the first one is
            // looked at (no branch around the loads: entries past the window hold stale parameters, whose clamped cell is in range
            // all the same), then the exit parameter and skip index of the empty ones from the packed cell.  The kernel shares
            // a SIMD with three other rays' waves and this phase is issue-bound: ~55 instead of ~105 vector instructions per probe
            // (eight probes per batch: sixteen would hold 168 VGPRs — three waves per SIMD, i.e. the 4,096 rays of a batch in two
            //  rounds instead of one)
            constexpr uint32_t kBatch = 8;
            const float mip_bound = fminf(1.0f, p.bound), mip_rbound = 1 / mip_bound;
            uint32_t prev_last = 0xffffffffu;  // (wave-uniform) cell of entry 64 i - 1
#pragma unroll 1
            for (uint32_t b0 = 0; b0 * 64 < wn; b0 += kBatch) {  // (uniform trip count; not unrolled: the batches must not merge)
                uint32_t bits[kBatch], idx[kBatch], cell[kBatch];
#pragma unroll
                for (uint32_t j = 0; j < kBatch; j++) {
                    const float t = T[lane + 64 * (b0 + j)];
                    const float x = clampf(__builtin_fmaf(t, r.dx, r.ox), -p.bound, p.bound);
                    const float y = clampf(__builtin_fmaf(t, r.dy, r.oy), -p.bound, p.bound);
                    const float z = clampf(__builtin_fmaf(t, r.dz, r.oz), -p.bound, p.bound);
                    const uint32_t nx = (uint32_t)(int)clampf(__builtin_fmaf(x, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
                    const uint32_t ny = (uint32_t)(int)clampf(__builtin_fmaf(y, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
                    const uint32_t nz = (uint32_t)(int)clampf(__builtin_fmaf(z, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
                    cell[j] = nx | (ny << 8) | (nz << 16);
                    idx[j] = (uint32_t)(float)(lut[nx] | (lut[ny] << 1) | (lut[nz] << 2));  // raymarching.cu:372: 0 * H3 + (float)morton
                }
#pragma unroll
                for (uint32_t j = 0; j < kBatch; j++) bits[j] = (uint32_t)p.grid[idx[j] >> 3];
#pragma unroll
                for (uint32_t j = 0; j < kBatch; j++) {
                    const uint32_t i = b0 + j, k = lane + 64 * i;
                    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)cell[j], (int)cell[j], 0x138, 0xF, 0xF, false);  // wave_shr:1
                    if (lane == 0) prev = prev_last;
                    prev_last = (uint32_t)__builtin_amdgcn_readlane((int)cell[j], 63);
                    if (cell[j] != prev) headm |= 1u << i;
                    if (k < wn) {
                        uint16_t e = (uint16_t)(k + 1);
                        if (bits[j] & (1u << (idx[j] & 7u))) {
                            occm |= 1u << i;
                        } else {
                            const float tk = T[k];
                            ProbeCell c;
                            c.x = clampf(__builtin_fmaf(tk, r.dx, r.ox), -p.bound, p.bound);
                            c.y = clampf(__builtin_fmaf(tk, r.dy, r.oy), -p.bound, p.bound);
                            c.z = clampf(__builtin_fmaf(tk, r.dz, r.oz), -p.bound, p.bound);
                            c.mip_bound = mip_bound;
                            c.nx = (int)(cell[j] & 255u); c.ny = (int)((cell[j] >> 8) & 255u); c.nz = (int)(cell[j] >> 16);
                            e = skip_index(k, tk, probe_exit(r, p, tk, c));
                        }
                        J[0][k] = e;
                    }
                }
            }
        } else {
            for (uint32_t k = lane; k < wn; k += 64) {
                float x, y, z, dt, tt;
                uint16_t e = (uint16_t)(k + 1);
                const float tk = T[k];
                if (!probe(r, p, tk, x, y, z, dt, tt)) e = skip_index(k, tk, tt);
                else occm |= 1u << (k >> 6);
                J[0][k] = e;
            }
        }
        S3D_TICK(1)
        if (lane == 0) J[0][wn] = (uint16_t)wn;
        mk[lane] = 0;
        // start of the walk: the first t_k at or past the skip target carried over from the previous window
        uint32_t start = wn;
        for (uint32_t k = lane; k < wn; k += 64)
            if (!(T[k] < pending_tt)) { start = k; break; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) start = min(start, (uint32_t)__shfl_xor((int)start, d, 64));
        pending_tt = -INFINITY;
        __syncthreads();
        // ---- C: mark the visited nodes.  Shortcut (FAST): the entries of one voxel crossing are consecutive, an empty voxel's
        // first entry jumps to the first entry of the next crossing, and inside an occupied voxel every entry is visited — so
        // the walk visits { k >= start : occupied(k) or first-of-its-voxel(k) } (plus the start itself) whenever every skip
        // target lands exactly on a voxel's first entry.  That set costs sixteen compares; it is ACCEPTED only if it is closed:
        // the jump targets of its members (known from phase B) must be exactly the set minus the start — with J[k] > k this
        // makes it the orbit of the start (each member's predecessor is a smaller member).  A target that lands one entry off
        // a voxel boundary (rounding of the exit parameter against the cell arithmetic) fails the test and takes the
        // pointer-doubling rounds below: same marks either way.
        bool marked = false;
#ifdef S3D_MARCH_PROFILE
        uint32_t dbg_vm0 = 0;
#endif
        if constexpr (FAST) {
            uint32_t vm = 0;
#pragma unroll
            for (uint32_t i = 0; i < kWin / 64; i++) {
                const uint32_t k = lane + 64 * i;
                const bool in = k < wn && k >= start;
                if (in && ((((occm | headm) >> i) & 1u) || k == start)) vm |= 1u << i;
            }
            // The candidate V is the orbit of the start for ~199 rays in 200.  It is ACCEPTED when it is closed (J(V) inside V) and
            // supported (every member but the start is some member's target): with J[k] > k that makes it exactly the orbit (the
            // smallest member is the start, every other member hangs off a smaller one).  Otherwise a skip target landed one entry
            // off a voxel boundary: an early landing adds a member the candidate lacks (not closed: V <- V + J(V)), a late one
            // leaves a voxel's first entry unvisited (closed but unsupported: V <- {start} + J(V) drops it) — a pass or two;
            // not settled after kMaxPasses: the pointer-doubling rounds below.
            const uint32_t startbit = (start < wn && (start & 63u) == lane) ? (1u << (start >> 6)) : 0u;
#ifdef S3D_MARCH_PROFILE
            dbg_vm0 = vm;
#endif
            constexpr uint32_t kMaxPasses = 6;
            for (uint32_t pass = 0; pass < kMaxPasses && !marked; pass++) {
#pragma unroll
                for (uint32_t i = 0; i < kWin / 64; i++) {
                    if ((vm >> i) & 1u) {
                        const uint32_t tg = J[0][lane + 64 * i];  // (written by this lane in phase B)
                        if (tg < wn) atomicOr(&mk[tg & 63], 1u << (tg >> 6));
                    }
                }
                __syncthreads();
                const uint32_t pm = mk[lane];
                const bool closed = __ballot((pm & ~vm) != 0u) == 0ull;
                const bool supported = __ballot((vm & ~startbit & ~pm) != 0u) == 0ull;
                marked = closed && supported;
                if (!marked) vm = closed ? (pm | startbit) : (vm | pm);
                __syncthreads();  // (everyone has read the marks before they are replaced)
                mk[lane] = marked ? vm : 0u;
                __syncthreads();
            }
        }
        if (!marked && lane == 0 && start < wn) mk[start & 63] = 1u << (start >> 6);
        __syncthreads();
        // pointer doubling (always for the general kernel; for FAST only when the shortcut's set was not closed)
        uint32_t cur = 0;
        constexpr uint32_t kPer = kWin / 64 + 1;    // entries per lane (k = lane + 64 i)
        const uint32_t iters = wn / 64 + 1;          // ... of which this window uses the first `iters` (covers k <= wn)
        for (uint32_t span = 1; span < wn && !marked; span <<= 1) {
            const uint16_t* Jc = J[cur];
            uint16_t* Jn = J[cur ^ 1];
            uint32_t j1[kPer], j2[kPer];
            const uint32_t mw = mk[lane];
#pragma unroll
            for (uint32_t i = 0; i < kPer; i++)
                if (i < iters) { const uint32_t k = lane + 64 * i; j1[i] = k <= wn ? Jc[k] : wn; }
#pragma unroll
            for (uint32_t i = 0; i < kPer; i++)
                if (i < iters) j2[i] = Jc[j1[i]];
#pragma unroll
            for (uint32_t i = 0; i < kPer; i++)
                if (i < iters) {
                    const uint32_t k = lane + 64 * i;
                    if (((mw >> i) & 1u) && j1[i] < wn) atomicOr(&mk[j1[i] & 63], 1u << (j1[i] >> 6));
                    if (k <= wn) Jn[k] = (uint16_t)j2[i];
                }
            cur ^= 1;
            __syncthreads();
            // the doubled jump from the start already leaves the window: every node of the walk is marked
            if (start >= wn || J[cur][start] >= wn) break;
        }
        S3D_TICK(2)
#ifdef S3D_MARCH_PROFILE
        if (FAST && max_steps >= 256) {  // candidate of the shortcut against the marks that were accepted: size and place of the difference
            const uint32_t d = mk[lane] ^ dbg_vm0;
            uint32_t cnt = __popc(d), first = d ? (lane + 64u * (uint32_t)__builtin_ctz(d)) : 0xFFFFu;
#pragma unroll
            for (int s2 = 32; s2 >= 1; s2 >>= 1) { cnt += (uint32_t)__shfl_xor((int)cnt, s2, 64); first = min(first, (uint32_t)__shfl_xor((int)first, s2, 64)); }
            if (lane == 0) { tout[max_steps - 16] = (float)cnt; tout[max_steps - 15] = (float)first; tout[max_steps - 14] = (float)wn; tout[max_steps - 13] = (float)start; }
        }
#endif
        // ---- emit the occupied visited samples in order; remember the last visited empty probe
        uint32_t last_empty = kWin;  // per lane, then wave max
        bool any_empty = false;
        for (uint32_t k0 = 0; k0 < wn && num_steps < max_steps; k0 += 64) {
            const uint32_t k = k0 + lane;
            const bool vis = k < wn && ((mk[lane] >> (k0 >> 6)) & 1u);
            const bool emit = vis && ((occm >> (k0 >> 6)) & 1u);
            if (vis && !emit) { last_empty = k; any_empty = true; }
            const unsigned long long m = __ballot(emit);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (emit && num_steps + rank < max_steps) tout[num_steps + rank] = T[k];
            num_steps = min(max_steps, num_steps + (uint32_t)__popcll(m));
        }
        // a visited empty probe whose skip target lies beyond a FULL window: recompute the target for the carry
        uint32_t le = any_empty ? last_empty : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) le = max(le, (uint32_t)__shfl_xor((int)le, d, 64));
        const bool had_empty = __ballot(any_empty) != 0;
        const bool more = (wn == kWin) && (num_steps < max_steps);
        if (more && had_empty) {
            // (its skip target index was wn  <=>  no later entry of the window reaches the target: T is increasing)
            float x, y, z, dt, tt;
            (void)probe(r, p, T[le], x, y, z, dt, tt);
            if (le + 1 >= wn || T[wn - 1] < tt) pending_tt = tt;
        }
        t_carry = T[kWin];  // only meaningful when the window was full
        __syncthreads();
        S3D_TICK(3)
        if (!more) break;
    }
#ifdef S3D_MARCH_PROFILE
    if (lane == 0 && max_steps >= 256) {  // phase times (100 MHz ticks) in the unused tail of the scratch row
        for (int i = 0; i < 4; i++) tout[max_steps - 8 + i] = (float)tp[i];
    }
#endif
    if (lane == 0) {
        rays[n * 3] = (int32_t)n;
        rays[n * 3 + 2] = (int32_t)num_steps;
        ws[kWsHeader + n] = num_steps;
        if (n == 0) ws[0] = (uint32_t)counter[0];
    }
}

__global__ void __launch_bounds__(64) k_march_write_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C,
                                                         uint32_t H, uint32_t M, const float* __restrict__ nears,
                                                         const float* __restrict__ noises, float* __restrict__ xyzs,
                                                         float* __restrict__ dirs, float* __restrict__ deltas,
                                                         int32_t* __restrict__ rays, int32_t* __restrict__ counter,
                                                         const uint32_t* __restrict__ ws) {
    const uint32_t n = blockIdx.x, lane = threadIdx.x;
    // offset of this ray's span = the counts of all earlier rays: every workgroup sums them itself (a scan kernel in between
    // would cost a launch, a chained scan a dependency across 4,096 workgroups).  Sixteen bytes per lane and trip, four
    // independent partial sums (the counts sit 16-byte aligned behind the 4-word header): up to 16 trips instead of 64
    uint32_t part = 0;
    {
        const uint4* c4 = reinterpret_cast<const uint4*>(ws + kWsHeader);
        const uint32_t n4 = n / 4;
        uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        for (uint32_t i = lane; i < n4; i += 64) {
            const uint4 v = c4[i];
            p0 += v.x; p1 += v.y; p2 += v.z; p3 += v.w;
        }
        part = (p0 + p1) + (p2 + p3);
        const uint32_t r0 = n4 * 4 + lane;
        if (r0 < n) part += ws[kWsHeader + r0];  // (n % 4 < 4 <= 64 lanes)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    const uint32_t off = ws[0] + part;
    const uint32_t num_steps = ws[kWsHeader + n];
    if (lane == 0) {
        rays[n * 3 + 1] = (int32_t)off;
        if (n == N - 1) { counter[0] = (int32_t)(off + num_steps); counter[1] = counter[1] + (int32_t)N; }
    }
    if (n == N - 1 && off + num_steps < M) zero_sample_rows(xyzs, dirs, deltas, off + num_steps, pad_end(off + num_steps, M), lane, 64);
    if (num_steps != 0 && off < M && off + num_steps > M) zero_sample_rows(xyzs, dirs, deltas, off, M, lane, 64);
    if (num_steps == 0 || off + num_steps > M) return;
    const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, nullptr);
    const Ray r = load_ray(rays_o, rays_d, n);
    float t0 = nears[n];
    t0 = __builtin_fmaf(clampf(t0 * dt_gamma, p.dt_min, p.dt_max), noises[n], t0);
    const float* tin = reinterpret_cast<const float*>(ws + kWsHeader + N) + (size_t)n * max_steps;
    for (uint32_t i = lane; i < num_steps; i += 64) {
        const float t = tin[i];
        float last_t = t0;
        if (i > 0) { const float tp = tin[i - 1]; last_t = tp + clampf(tp * dt_gamma, p.dt_min, p.dt_max); }
        const float dt = clampf(t * dt_gamma, p.dt_min, p.dt_max);
        const size_t o = (size_t)off + i;
        xyzs[o * 3] = clampf(__builtin_fmaf(t, r.dx, r.ox), -bound, bound);
        xyzs[o * 3 + 1] = clampf(__builtin_fmaf(t, r.dy, r.oy), -bound, bound);
        xyzs[o * 3 + 2] = clampf(__builtin_fmaf(t, r.dz, r.oz), -bound, bound);
        dirs[o * 3] = r.dx; dirs[o * 3 + 1] = r.dy; dirs[o * 3 + 2] = r.dz;
        deltas[o * 2] = dt;
        deltas[o * 2 + 1] = (t + dt) - last_t;
    }
}

constexpr uint32_t kWaveMarchMaxRays = 16384;

// ---------------------------------------------------------------- composite (training)
__global__ void __launch_bounds__(64) k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                            const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                            uint32_t M, uint32_t N, float T_thresh,
                                                            float* __restrict__ weights_sum, float* __restrict__ depth,
                                                            float* __restrict__ image) {
    const uint32_t n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (num_steps != 0 && offset + num_steps <= M) {
        const float* s = sigmas + offset;
        const float* c = rgbs + (size_t)offset * 3;
        const float2* dl = reinterpret_cast<const float2*>(deltas) + offset;
        float T = 1.0f, t = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float2 dd = dl[step];
            const float alpha = 1.0f - __expf(-s[step] * dd.x);
            const float weight = alpha * T;
            r = __builtin_fmaf(weight, c[step * 3], r);
            g = __builtin_fmaf(weight, c[step * 3 + 1], g);
            b = __builtin_fmaf(weight, c[step * 3 + 2], b);
            t += dd.y;
            d = __builtin_fmaf(weight, t, d);
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
        }
    }
    weights_sum[index] = ws; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

__global__ void __launch_bounds__(64) k_composite_train_bwd(const float* __restrict__ grad_weights_sum,
                                                            const float* __restrict__ grad_image,
                                                            const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                            const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                            const float* __restrict__ weights_sum,
                                                            const float* __restrict__ image, uint32_t M, uint32_t N,
                                                            float T_thresh, float* __restrict__ grad_sigmas,
                                                            float* __restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (n == N - 1 && offset + num_steps < M) zero_grad_rows(grad_sigmas, grad_rgbs, offset + num_steps, pad_end(offset + num_steps, M), 0, 1);
    if (num_steps != 0 && offset < M && offset + num_steps > M) zero_grad_rows(grad_sigmas, grad_rgbs, offset, M, 0, 1);
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index];
    const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
    const float ws_final = weights_sum[index];
    const float* s = sigmas + offset;
    const float* c = rgbs + (size_t)offset * 3;
    const float2* dl = reinterpret_cast<const float2*>(deltas) + offset;
    float* gs = grad_sigmas + offset;
    float* gc = grad_rgbs + (size_t)offset * 3;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
    for (uint32_t step = 0; step < num_steps; step++) {
        const float2 dd = dl[step];
        const float c0 = c[step * 3], c1 = c[step * 3 + 1], c2 = c[step * 3 + 2];
        const float alpha = 1.0f - __expf(-s[step] * dd.x);
        const float weight = alpha * T;
        r = __builtin_fmaf(weight, c0, r); g = __builtin_fmaf(weight, c1, g); b = __builtin_fmaf(weight, c2, b);
        ws += weight;
        T *= 1.0f - alpha;
        gc[step * 3] = gi0 * weight; gc[step * 3 + 1] = gi1 * weight; gc[step * 3 + 2] = gi2 * weight;
        float acc = gi0 * __builtin_fmaf(T, c0, -(r_final - r));
        acc = __builtin_fmaf(gi1, __builtin_fmaf(T, c1, -(g_final - g)), acc);
        acc = __builtin_fmaf(gi2, __builtin_fmaf(T, c2, -(b_final - b)), acc);
        acc = __builtin_fmaf(gws, 1 - ws_final, acc);
        gs[step] = dd.x * acc;
        if (T < T_thresh) {  // samples behind the termination take no gradient
            zero_grad_rows(grad_sigmas, grad_rgbs, offset + step + 1, offset + num_steps, 0, 1);
            break;
        }
    }
}


// ---------------------------------------------------------------- composite (training), one WAVE per ray
// The lane-per-ray kernels above walk each span serially with uncoalesced loads (4,096 rays = 64 waves, ~125 us).
// Here a wave owns one ray: 64 consecutive samples are loaded coalesced, transmittance is a wave prefix PRODUCT,
// depth a wave prefix SUM, early termination a prefix property (sample i is used iff the transmittance in front of
// it is still >= T_thresh), and the pixel is a wave reduction.  Same formulas as raymarching.cu:540-567 / 643-681;
// the products/sums are re-associated (tree instead of chain), covered by the FP tolerance of the parity tests.
__device__ __forceinline__ float wave_scan_mul(float v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_up(v, d, 64); if (lane >= (uint32_t)d) v *= o; }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_up(v, d, 64); if (lane >= (uint32_t)d) v += o; }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct RayPixel { float r, g, b, ws, depth; };
// forward of one ray by one wave; every lane returns the same (butterfly-summed) pixel
__device__ __forceinline__ RayPixel composite_ray_fwd_wave(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                          const float* __restrict__ deltas, uint32_t offset, uint32_t num_steps,
                                                          uint32_t M, float T_thresh, uint32_t lane) {
    float r = 0, g = 0, b = 0, ws = 0, dsum = 0;
    if (num_steps != 0 && offset + num_steps <= M) {
        float T_carry = 1.0f, t_carry = 0.0f;
        for (uint32_t base = 0; base < num_steps; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < num_steps;
            const size_t o = (size_t)offset + i;
            const float sg = valid ? sigmas[o] : 0.0f;
            const float2 dd = valid ? reinterpret_cast<const float2*>(deltas)[o] : make_float2(0.0f, 0.0f);
            const float c0 = valid ? rgbs[o * 3] : 0.0f, c1 = valid ? rgbs[o * 3 + 1] : 0.0f, c2 = valid ? rgbs[o * 3 + 2] : 0.0f;
            const float alpha = 1.0f - __expf(-sg * dd.x);
            const float P = wave_scan_mul(1.0f - alpha, lane);
            float Pex = __shfl_up(P, 1, 64);
            if (lane == 0) Pex = 1.0f;
            const float T = T_carry * Pex;                       // transmittance in front of sample i
            const float tc = t_carry + wave_scan_add(dd.y, lane);  // ray parameter after sample i
            const bool used = valid && (i == 0 || !(T < T_thresh));
            const float w = used ? alpha * T : 0.0f;
            r += w * c0; g += w * c1; b += w * c2; ws += w; dsum += w * tc;
            T_carry *= __shfl(P, 63, 64);
            t_carry = __shfl(tc, 63, 64);
            if (T_carry < T_thresh) break;
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); dsum = wave_sum(dsum);
    }
    return RayPixel{r, g, b, ws, dsum};
}

__global__ void __launch_bounds__(256) k_composite_train_fwd_wave(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                 const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                 uint32_t M, uint32_t N, float T_thresh,
                                                                 float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                 float* __restrict__ image) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    const RayPixel px = composite_ray_fwd_wave(sigmas, rgbs, deltas, offset, num_steps, M, T_thresh, lane);
    if (lane == 0) {
        weights_sum[index] = px.ws; depth[index] = px.depth;
        image[index * 3] = px.r; image[index * 3 + 1] = px.g; image[index * 3 + 2] = px.b;
    }
}

// backward of ray n by one wave (gws, gi*: the gradient of its pixel; rf, gf, bf, wsf: the pixel)
__device__ __forceinline__ void composite_ray_bwd_wave(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                       const float* __restrict__ deltas, uint32_t n, uint32_t offset,
                                                       uint32_t num_steps, uint32_t M, uint32_t N, float T_thresh, float gws,
                                                       float gi0, float gi1, float gi2, float rf, float gf, float bf, float wsf,
                                                       float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs,
                                                       uint32_t lane) {
    if (n == N - 1 && offset + num_steps < M) zero_grad_rows(grad_sigmas, grad_rgbs, offset + num_steps, pad_end(offset + num_steps, M), lane, 64);
    if (num_steps != 0 && offset < M && offset + num_steps > M) zero_grad_rows(grad_sigmas, grad_rgbs, offset, M, lane, 64);
    if (num_steps == 0 || offset + num_steps > M) return;
    float T_carry = 1.0f, rc = 0, gc = 0, bc = 0;  // running composites up to the previous chunk
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < num_steps;
        const size_t o = (size_t)offset + i;
        const float sg = valid ? sigmas[o] : 0.0f;
        const float d0 = valid ? deltas[o * 2] : 0.0f;
        const float c0 = valid ? rgbs[o * 3] : 0.0f, c1 = valid ? rgbs[o * 3 + 1] : 0.0f, c2 = valid ? rgbs[o * 3 + 2] : 0.0f;
        const float alpha = 1.0f - __expf(-sg * d0);
        const float P = wave_scan_mul(1.0f - alpha, lane);
        float Pex = __shfl_up(P, 1, 64);
        if (lane == 0) Pex = 1.0f;
        const float T = T_carry * Pex;   // in front of sample i
        const float Tn = T_carry * P;    // behind sample i
        const bool used = valid && (i == 0 || !(T < T_thresh));
        const float w = used ? alpha * T : 0.0f;
        const float rr = rc + wave_scan_add(w * c0, lane), gg = gc + wave_scan_add(w * c1, lane), bb = bc + wave_scan_add(w * c2, lane);
        if (used) {
            grad_rgbs[o * 3] = gi0 * w; grad_rgbs[o * 3 + 1] = gi1 * w; grad_rgbs[o * 3 + 2] = gi2 * w;
            float acc = gi0 * (Tn * c0 - (rf - rr));
            acc += gi1 * (Tn * c1 - (gf - gg));
            acc += gi2 * (Tn * c2 - (bf - bb));
            acc += gws * (1 - wsf);
            grad_sigmas[o] = d0 * acc;
        } else if (valid) {  // behind the termination: no gradient
            grad_rgbs[o * 3] = 0.0f; grad_rgbs[o * 3 + 1] = 0.0f; grad_rgbs[o * 3 + 2] = 0.0f;
            grad_sigmas[o] = 0.0f;
        }
        T_carry *= __shfl(P, 63, 64);
        rc = __shfl(rr, 63, 64); gc = __shfl(gg, 63, 64); bc = __shfl(bb, 63, 64);
        if (T_carry < T_thresh) {
            zero_grad_rows(grad_sigmas, grad_rgbs, offset + base + 64, offset + num_steps, lane, 64);
            break;
        }
    }
}

__global__ void __launch_bounds__(256) k_composite_train_bwd_wave(const float* __restrict__ grad_weights_sum,
                                                                 const float* __restrict__ grad_image,
                                                                 const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                 const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                                 const float* __restrict__ weights_sum,
                                                                 const float* __restrict__ image, uint32_t M, uint32_t N,
                                                                 float T_thresh, float* __restrict__ grad_sigmas,
                                                                 float* __restrict__ grad_rgbs) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    composite_ray_bwd_wave(sigmas, rgbs, deltas, n, offset, num_steps, M, N, T_thresh, grad_weights_sum[index], grad_image[index * 3],
                           grad_image[index * 3 + 1], grad_image[index * 3 + 2], image[index * 3], image[index * 3 + 1],
                           image[index * 3 + 2], weights_sum[index], grad_sigmas, grad_rgbs, lane);
}

// ---- compositing, background + MSE criterion and the compositing backward of one training ray batch in ONE launch
// (k_composite_train_fwd_wave -> ngp_head.hip:k_bg_mse_forward with its announced upstream gradient -> k_composite_train_bwd_wave:
// three launches of ~8 us each for 4,096 rays, all three bound by their launch and one pass of latencies).  The gradient of a mean
// of squared errors w.r.t. a ray's pixel needs that ray alone, so the wave that composited a ray goes straight on to its
// backward (its samples are in L2); only the loss VALUE needs every ray: each wave files its three squared errors (and the depth
// term) and a one-workgroup launch behind this one adds them in k_bg_mse_forward's order — thread t of 1,024 takes rays t,
// t + 1,024, ..., butterfly per 64, sixteen partials in sequence — so loss, pixel and gradients equal the three-launch sequence
// bit for bit.  (Measured first with the sum inside this kernel, by the last workgroup to take a ticket: 80 us instead of ~12 —
// 1,024 workgroups each paying an agent-scope release, i.e. an L2 write-back, and queueing on one ticket word.)
// work: [3N squared errors | N depth terms]
__global__ void __launch_bounds__(256) k_composite_train_loss_wave(
    const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas, const int32_t* __restrict__ rays,
    uint32_t M, uint32_t N, float T_thresh, const float* __restrict__ gt, float bg0, float bg1, float bg2,
    const float* __restrict__ grad_loss, const float* __restrict__ gt_depth, float depth_weight, float* __restrict__ weights_sum,
    float* __restrict__ depth, float* __restrict__ image, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs,
    float* __restrict__ grad_image, float* __restrict__ grad_ws, float* __restrict__ work) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    float* sq = work;
    float* dabs = work + (size_t)3 * N;
    if (n >= N) return;
    {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        const RayPixel px = composite_ray_fwd_wave(sigmas, rgbs, deltas, offset, num_steps, M, T_thresh, lane);
        // ngp_head.hip:k_bg_mse_forward for this ray
        const float bg[3] = {bg0, bg1, bg2};
        const float pix[3] = {px.r, px.g, px.b};
        const float k = *grad_loss * (2.0f / (3.0f * (float)N));
        const float w = 1.0f - px.ws;
        float gd[3], gw = 0.0f, dd[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = (pix[c] + w * bg[c]) - gt[(size_t)index * 3 + c];
            dd[c] = d * d;
            gd[c] = k * d;
            gw -= gd[c] * bg[c];
        }
        if (lane == 0) {
            weights_sum[index] = px.ws; depth[index] = px.depth;
            image[index * 3] = px.r; image[index * 3 + 1] = px.g; image[index * 3 + 2] = px.b;
            sq[(size_t)index * 3] = dd[0]; sq[(size_t)index * 3 + 1] = dd[1]; sq[(size_t)index * 3 + 2] = dd[2];
            if (gt_depth) {
                float dv = px.depth;
                dv = dv != dv ? 0.0f : fminf(fmaxf(dv, -3.402823466e38f), 3.402823466e38f);  // torch.nan_to_num(nan=0.)
                dabs[index] = fabsf(dv - gt_depth[index]);
            }
            if (grad_image) {
                grad_image[(size_t)index * 3] = gd[0]; grad_image[(size_t)index * 3 + 1] = gd[1]; grad_image[(size_t)index * 3 + 2] = gd[2];
                grad_ws[index] = gw;
            }
        }
        composite_ray_bwd_wave(sigmas, rgbs, deltas, n, offset, num_steps, M, N, T_thresh, gw, gd[0], gd[1], gd[2], px.r, px.g, px.b,
                               px.ws, grad_sigmas, grad_rgbs, lane);
    }
}

// the value of the criterion from the terms k_composite_train_loss_wave filed, in k_bg_mse_forward's order of additions (one
// workgroup, thread t takes rays t, t + 1,024, ...)
__global__ void __launch_bounds__(1024) k_bg_mse_reduce(const float* __restrict__ sq, const float* __restrict__ dabs, uint32_t N,
                                                        float depth_weight, float* __restrict__ loss) {
    __shared__ float part[16], dpart[16];
    float acc = 0.0f, dacc = 0.0f;
    for (uint32_t m = threadIdx.x; m < N; m += 1024) {
#pragma unroll
        for (int c = 0; c < 3; c++) acc += sq[(size_t)m * 3 + c];
    }
    if (dabs)
        for (uint32_t m = threadIdx.x; m < N; m += 1024) dacc += dabs[m];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { acc += __shfl_xor(acc, d, 64); dacc += __shfl_xor(dacc, d, 64); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = acc; dpart[threadIdx.x >> 6] = dacc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f, td = 0.0f;
        for (int w = 0; w < 16; w++) { t += part[w]; td += dpart[w]; }
        t = t / (3.0f * (float)N);
        if (dabs) t = t + depth_weight * (td / (float)N);
        *loss = t;
    }
}

// ---------------------------------------------------------------- inference
__device__ __forceinline__ uint32_t alive_count(uint32_t bound, const int32_t* __restrict__ dev) {
    if (!dev) return bound;
    const int32_t v = *dev;
    return v <= 0 ? 0u : ((uint32_t)v < bound ? (uint32_t)v : bound);
}

// ---- march_rays, second generation: two launches ----
// Measured on the first one (r08, 800x800 frame, tools/run_pmc_render.sh): 6,400 vector instructions per wave at 35 % active
// lanes — a ray ends after n_step samples or at its far plane, its lane idles until the slowest of the wave's 64 rays is
// done — 12 quarter-rate integer multiplies per probe in the morton code, and 32 bytes per sample leaving in 4-byte stores
// 32 * n_step bytes apart (186 MB of HBM writes per launch for 80 MB of samples).  Now:
//  (1) k_march_rays_t walks the rays and records, per emitted sample, only (t, previous t) into the sample's own `deltas` row.
//      A wave owns a POOL of consecutive ray slots (64 ... 256) and hands the next one to a lane that has finished its ray
//      (ballot + prefix count, no atomics): lanes stay busy until the pool runs dry.  Morton codes come from a per-workgroup
//      LDS table of spread coordinates; single-cascade grids (bound <= 1) skip the mip-level arithmetic.
//  (2) k_march_rays_expand, one lane per sample row, turns (t, previous t) into xyz / dir / (dt, t' - previous t) with the
//      reference's expressions (raymarching.cu:750-785) and writes whole rows side by side; slots a ray did not fill and the
//      padding rows are zeroed here.
// Same samples, bit for bit (the t sequence and every emitted value are computed by the same float expressions).
template <bool C1>
__device__ __forceinline__ bool probe_lut(const Ray& r, const MarchParams& p, const uint32_t* __restrict__ lut, float t,
                                          float& dt, float& t_skip) {
    const float x = clampf(__builtin_fmaf(t, r.dx, r.ox), -p.bound, p.bound);
    const float y = clampf(__builtin_fmaf(t, r.dy, r.oy), -p.bound, p.bound);
    const float z = clampf(__builtin_fmaf(t, r.dz, r.oz), -p.bound, p.bound);
    dt = clampf(t * p.dt_gamma, p.dt_min, p.dt_max);
    int level = 0;
    float mip_bound, mip_rbound;
    if constexpr (C1) {  // one cascade: both exponents clamp to level 0 (mip_level with Cf = 1)
        mip_bound = fminf(1.0f, p.bound);
        mip_rbound = 1 / mip_bound;
    } else {
        level = mip_level(x, y, z, dt, p);
        mip_bound = fminf(ldexpf(1.0f, level), p.bound);
        mip_rbound = 1 / mip_bound;
    }
    const int nx = (int)clampf(__builtin_fmaf(x, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);  // (see probe(): no fp64 needed)
    const int ny = (int)clampf(__builtin_fmaf(y, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
    const int nz = (int)clampf(__builtin_fmaf(z, mip_rbound, 1.0f) * p.halfH, 0.0f, p.Hm1);
    const uint32_t m = lut[nx] | (lut[ny] << 1) | (lut[nz] << 2);
    uint32_t index;
    if constexpr (C1) index = (uint32_t)(float)m;  // raymarching.cu:769 forms `level * H3 + morton` in float: 0 * H3 + (float)m
    else index = (uint32_t)((float)level * p.H3 + (float)m);
    const bool occ = (p.grid[index >> 3] & (1u << (index & 7u))) != 0;
    if (!occ) {
        const float sx = copysignf(1.0f, r.dx), sy = copysignf(1.0f, r.dy), sz = copysignf(1.0f, r.dz);
        const float tx = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.5f, sx, (float)nx + 0.5f) * p.rH, 2.0f, -1.0f), mip_bound, -x) * r.rdx;
        const float ty = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.5f, sy, (float)ny + 0.5f) * p.rH, 2.0f, -1.0f), mip_bound, -y) * r.rdy;
        const float tz = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.5f, sz, (float)nz + 0.5f) * p.rH, 2.0f, -1.0f), mip_bound, -z) * r.rdz;
        t_skip = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return occ;
}

#ifndef S3D_MARCH_GROUP_MAX
#define S3D_MARCH_GROUP_MAX 320000
#endif
constexpr uint32_t kMarchGroupMaxRays = S3D_MARCH_GROUP_MAX;  // alive rays up to which the 16-lanes-per-ray walk is used (0 = never)
constexpr uint32_t kMarchRefill = 16;  // idle lanes of a wave before the next ray slots of its pool are handed out

template <bool C1>
__global__ void __launch_bounds__(256) k_march_rays_t(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                                                      const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                      const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                      uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                      const float* __restrict__ fars, float2* __restrict__ tl,
                                                      const float* __restrict__ noises, const int32_t* __restrict__ n_alive_dev,
                                                      int32_t* __restrict__ n_rows_out, uint32_t pool) {
    __shared__ uint32_t lut[1024];
    for (uint32_t i = threadIdx.x; i < H; i += 256) lut[i] = expand_bits(i);
    __syncthreads();
    const uint32_t live = alive_count(n_alive, n_alive_dev);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_rows_out) *n_rows_out = (int32_t)(live * n_step);
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t next = wave * pool;  // (wave-uniform) first slot of the pool not handed out yet
    const uint32_t pool_end = min(next + pool, live);
    if (next >= pool_end) return;
    const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, grid);
    Ray r{};
    float t = 0.0f, last_t = 0.0f, far = 0.0f;
    uint32_t step = 0;
    float2* row = tl;
    bool busy = false;
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (next < pool_end && (n_idle >= kMarchRefill || n_idle == 64u)) {
            if (!busy) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t n = next + rank;
                if (n < pool_end) {
                    const uint32_t index = (uint32_t)rays_alive[n];
                    r = load_ray(rays_o, rays_d, index);
                    t = rays_t[index];
                    far = fars[index];
                    t = __builtin_fmaf(clampf(t * dt_gamma, p.dt_min, p.dt_max), noises ? noises[n] : 0.0f, t);  // (no noise: t + x * 0 = t)
                    last_t = t;
                    step = 0;
                    row = tl + (size_t)n * n_step;
                    busy = true;
                }
            }
            next += n_idle;
        } else if (n_idle == 64u) break;  // nothing running, nothing left to hand out
        if (busy) {
            if (t < far && step < n_step) {
                float dt, tt;
                if (probe_lut<C1>(r, p, lut, t, dt, tt)) {
                    row[step] = make_float2(t, last_t);
                    t += dt;
                    last_t = t;
                    step++;
                } else t = skip_to(p, t, tt);
            } else {
                for (; step < n_step; step++) row[step] = make_float2(-1.0f, 0.0f);  // unfilled slot: (t < 0)
                busy = false;
            }
        }
    }
}

// ---- the t walk with SIXTEEN lanes per ray (later iterations of the loop: fewer rays, more steps each) ----
// With a lane per ray, 1.5e5 alive rays are 2.4 waves per SIMD, each a chain of ~40 dependent probes (110 us per launch at
// 13 us of vector work).  The walk only ever visits elements of ONE sequence t_0, t_1 = t_0 + dt(t_0), ... — an occupied
// sample advances by one element, an empty one to the first element at or behind its voxel exit (raymarching.cu:788-799) —
// so a group of 16 lanes probes 16 consecutive elements at once and then resolves which of them the reference's loop
// visits: next[j] = j + 1 (occupied) or the number of window elements in front of the exit (empty; binary search over the
// group's sorted t by ds_bpermute), the nodes on the path from the start by pointer doubling on 16-bit reach masks, the
// occupied visited ones are emitted in order.  A skip that leaves the window is carried as `pending` into the next one.
// Every cross-lane read is executed by all lanes of a group (a disabled lane reads as 0).
template <bool C1, bool G0>  // C1: one cascade; G0: dt_gamma == 0 (constant step: clamp(t * 0) = dt_min)
__global__ void __launch_bounds__(256) k_march_rays_g(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                                                      const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                      const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                      uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                      const float* __restrict__ fars, float2* __restrict__ tl,
                                                      const float* __restrict__ noises, const int32_t* __restrict__ n_alive_dev,
                                                      int32_t* __restrict__ n_rows_out) {
    __shared__ uint32_t lut[1024];
    for (uint32_t i = threadIdx.x; i < H; i += 256) lut[i] = expand_bits(i);
    __syncthreads();
    const uint32_t live = alive_count(n_alive, n_alive_dev);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_rows_out) *n_rows_out = (int32_t)(live * n_step);
    const uint32_t g = (blockIdx.x * 256 + threadIdx.x) >> 4;  // ray slot of this group
    const uint32_t j = threadIdx.x & 15u;
    if (g >= live) return;  // (whole groups leave)
    const uint32_t gbase = threadIdx.x & 48u;  // first lane of the group inside its wave
    const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, grid);
    const uint32_t index = (uint32_t)rays_alive[g];
    const Ray r = load_ray(rays_o, rays_d, index);
    const float far = fars[index];
    float t_base = rays_t[index];
    t_base = __builtin_fmaf(clampf(t_base * dt_gamma, p.dt_min, p.dt_max), noises ? noises[g] : 0.0f, t_base);
    float last_t = t_base;
    float pending = -1.0f;  // voxel exit of a skip that left the previous window (t is positive: -1 = none)
    uint32_t step = 0;
    float2* row = tl + (size_t)g * n_step;
    auto grp = [&](auto v, uint32_t lane_in_group) { return __shfl(v, (int)(gbase + lane_in_group), 64); };
    // elements of the window in front of x (0 .. 16); the window's t are ascending
    auto count_before = [&](float t, float x) {
        const float t15 = grp(t, 15u);
        uint32_t c = 0;
#pragma unroll
        for (uint32_t sft = 8; sft >= 1; sft >>= 1) {
            const float v = grp(t, c + sft - 1u);
            if (v < x) c += sft;
        }
        return t15 < x ? 16u : c;
    };
    for (;;) {
        // lane j holds element j of the window: j applications of t += dt(t) to the window's first element
        auto dt_of = [&](float tt) { return G0 ? clampf(0.0f, p.dt_min, p.dt_max) : clampf(tt * dt_gamma, p.dt_min, p.dt_max); };
        float t = t_base;
#pragma unroll
        for (uint32_t k = 0; k < 15; k++) {
            const float tn = t + dt_of(t);
            t = k < j ? tn : t;
        }
        const float t_next = t + dt_of(t);  // element j + 1
        const float t16 = grp(t_next, 15u);
        const bool valid = t < far;
        bool occ = false;
        float tskip = 0.0f, dtj;
        if (valid) occ = probe_lut<C1>(r, p, lut, t, dtj, tskip);
        const bool empty = valid && !occ;
        const unsigned long long wocc = __ballot(occ), wval = __ballot(valid);
        const uint32_t occm = (uint32_t)(wocc >> gbase) & 0xffffu, valm = (uint32_t)(wval >> gbase) & 0xffffu;
        // (the branches below are taken by whole groups: the cross-lane reads inside see all 16 lanes of their group)
        const uint32_t start = pending < 0.0f ? 0u : count_before(t, pending);
        uint32_t nxt = !valid ? 16u : j + 1u;  // node the reference's loop goes to from here
        uint32_t visited;
        if ((valm & ~occm) == 0u) {
            // no empty element in the window (inside an object): the walk takes every element from the start on
            visited = start < 16u ? (0xffffu << start) & 0xffffu : 0u;
            if (valm != 0xffffu) visited &= (valm << 1) | 1u;  // ... up to and including the first one behind the far plane
        } else {
            const uint32_t cb = count_before(t, empty ? tskip : t);
            if (empty) nxt = cb > j + 1u ? cb : j + 1u;
            // nodes on the path from each node, by pointer doubling (a path has at most 16 hops)
            uint32_t reach = 1u << j, jump = nxt;
#pragma unroll
            for (uint32_t rd = 0; rd < 4; rd++) {
                const uint32_t src = jump < 16u ? jump : 15u;
                const uint32_t rj = grp(reach, src), jj = grp(jump, src);
                if (jump < 16u) { reach |= rj; jump = jj; }
            }
            visited = start < 16u ? grp(reach, start) : 0u;
        }
        const uint32_t emit = visited & occm & valm;
        const uint32_t room = n_step - step;
        const uint32_t rank = (uint32_t)__popc(emit & ((1u << j) - 1u));
        const bool mine = ((emit >> j) & 1u) != 0u && rank < room;
        // previous emitted element of the window (highest emit bit below j): its t_next is this sample's `last_t`
        const uint32_t below = emit & ((1u << j) - 1u);
        const uint32_t prev = below ? 31u - (uint32_t)__clz(below) : 0u;
        const float lt_prev = grp(t_next, prev);
        if (mine) row[step + rank] = make_float2(t, below ? lt_prev : last_t);
        const uint32_t ne = min((uint32_t)__popc(emit), room);
        if (ne) {  // (group-uniform) the last emitted element's t_next is the new last_t
            const unsigned long long wl = __ballot(mine && rank == ne - 1u);
            const uint32_t q = (uint32_t)__ffsll((long long)((wl >> gbase) & 0xffffull)) - 1u;
            last_t = grp(t_next, q);
            step += ne;
        } else {
            (void)__ballot(false);
            (void)grp(t_next, 0u);
        }
        // the walk ends at n_step samples or at the first visited element behind the far plane
        const bool far_hit = (visited & ~valm) != 0u || valm == 0u;  // (a window entirely behind the far plane: so is whatever comes next)
        if (step >= n_step || far_hit) {
            for (uint32_t sidx = step + j; sidx < n_step; sidx += 16) row[sidx] = make_float2(-1.0f, 0.0f);  // unfilled slots: (t < 0)
            break;
        }
        // next window; a skip that pointed behind this one is carried
        const uint32_t lastv = visited ? 31u - (uint32_t)__clz(visited) : 0u;
        const float carry = grp(occ ? -1.0f : tskip, lastv);
        const uint32_t nl = grp(nxt, lastv);
        if (visited) pending = nl >= 16u ? carry : -1.0f;  // (no node visited: the whole window lies in front of `pending`)
        t_base = t16;
    }
}

__global__ void __launch_bounds__(256) k_march_rays_expand(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                                                           const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                           float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                                           float* __restrict__ xyzs, float* __restrict__ dirs,
                                                           float2* __restrict__ deltas, const int32_t* __restrict__ n_alive_dev,
                                                           uint32_t rows_total, bool zero_unfilled) {
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    const uint32_t live = alive_count(n_alive, n_alive_dev);
    const uint32_t live_rows = live * n_step;
    float x = 0.0f, y = 0.0f, z = 0.0f, dx = 0.0f, dy = 0.0f, dz = 0.0f, d0 = 0.0f, d1 = 0.0f;
    if (row >= live_rows) {
        // rows behind the live rays that a consumer may read — up to the next multiple of 128 of the live rows when the count is
        // on the device (n_valid), the whole buffer otherwise — are zeroed when the caller's buffers arrive uninitialised
        const uint32_t end = n_alive_dev ? min(rows_total, (live_rows + 127u) & ~127u) : rows_total;
        if (!zero_unfilled || row >= end) return;
    } else {
        const float2 v = deltas[row];
        if (v.x >= 0.0f) {
            const uint32_t n = row / n_step;
            const uint32_t index = (uint32_t)rays_alive[n];
            const MarchParams p = make_params(bound, dt_gamma, max_steps, C, H, nullptr);
            const float t = v.x;
            dx = rays_d[index * 3]; dy = rays_d[index * 3 + 1]; dz = rays_d[index * 3 + 2];
            x = clampf(__builtin_fmaf(t, dx, rays_o[index * 3]), -bound, bound);
            y = clampf(__builtin_fmaf(t, dy, rays_o[index * 3 + 1]), -bound, bound);
            z = clampf(__builtin_fmaf(t, dz, rays_o[index * 3 + 2]), -bound, bound);
            d0 = clampf(t * dt_gamma, p.dt_min, p.dt_max);
            d1 = (t + d0) - v.y;
        }
    }
    xyzs[(size_t)row * 3] = x; xyzs[(size_t)row * 3 + 1] = y; xyzs[(size_t)row * 3 + 2] = z;
    dirs[(size_t)row * 3] = dx; dirs[(size_t)row * 3 + 1] = dy; dirs[(size_t)row * 3 + 2] = dz;
    deltas[row] = make_float2(d0, d1);
}

// TS / TC: float or __half — the network's outputs are taken as they come (the reference's wrapper casts them to fp32 first,
// raymarching.py:356 custom_fwd; converting a binary16 on load gives the same value without the two cast launches per iteration)
template <typename TS, typename TC>
__global__ void __launch_bounds__(64) k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                                       int32_t* __restrict__ rays_alive, float* __restrict__ rays_t,
                                                       const TS* __restrict__ sigmas, const TC* __restrict__ rgbs,
                                                       const float* __restrict__ deltas, float* __restrict__ weights_sum,
                                                       float* __restrict__ depth, float* __restrict__ image,
                                                       const int32_t* __restrict__ n_alive_dev, bool vec4) {
    const uint32_t n = blockIdx.x * 64 + threadIdx.x;
    if (n >= alive_count(n_alive, n_alive_dev)) return;
    const uint32_t index = (uint32_t)rays_alive[n];
    const TS* s = sigmas + (size_t)n * n_step;
    const TC* c = rgbs + (size_t)n * n_step * 3;
    const float2* dl = reinterpret_cast<const float2*>(deltas) + (size_t)n * n_step;
    float t = rays_t[index];
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    // one sample: returns true when the ray's chunk ends here (raymarching.cu:848-889: the order of the tests and updates)
    auto sample = [&](float sg, float c0, float c1, float c2, float2 dd) {
        if (dd.x == 0) return true;
        const float alpha = 1.0f - __expf(-sg * dd.x);
        const float T = 1 - weight_sum;
        const float weight = alpha * T;
        weight_sum += weight;
        t += dd.y;
        d = __builtin_fmaf(weight, t, d);
        r = __builtin_fmaf(weight, c0, r);
        g = __builtin_fmaf(weight, c1, g);
        b = __builtin_fmaf(weight, c2, b);
        if (T < T_thresh) return true;
        step++;
        return false;
    };
    if (vec4) {
        // four samples per trip through 16-byte (fp16: 8-byte) loads: a lane's rows are contiguous but 64 lanes are 64 different
        // cache lines per load instruction — the kernel was bound by the L1's line rate (5 loads per sample and wave), not by
        // bytes; same arithmetic in the same order.  fp32 rows need only their natural 4-byte alignment (global memory takes
        // misaligned multi-dword accesses), so any n_step works; fp16 rows are vectorised when the launcher found them aligned.
        struct alignas(sizeof(TS) == 4 ? 4 : 8) S4 { TS v[4]; };
        struct alignas(sizeof(TC) == 4 ? 4 : 8) C4 { TC v[4]; };
        struct alignas(4) D4 { float x, y, z, w; };
        bool stop = false;
        uint32_t s0 = 0;
        for (; s0 + 4 <= n_step && !stop; s0 += 4) {
            const S4 sv = *reinterpret_cast<const S4*>(s + s0);
            const C4 ca = *reinterpret_cast<const C4*>(c + s0 * 3), cb = *reinterpret_cast<const C4*>(c + s0 * 3 + 4),
                     cc = *reinterpret_cast<const C4*>(c + s0 * 3 + 8);
            const D4 da = *reinterpret_cast<const D4*>(dl + s0), db = *reinterpret_cast<const D4*>(dl + s0 + 2);
            stop = sample((float)sv.v[0], (float)ca.v[0], (float)ca.v[1], (float)ca.v[2], make_float2(da.x, da.y)) ||
                   sample((float)sv.v[1], (float)ca.v[3], (float)cb.v[0], (float)cb.v[1], make_float2(da.z, da.w)) ||
                   sample((float)sv.v[2], (float)cb.v[2], (float)cb.v[3], (float)cc.v[0], make_float2(db.x, db.y)) ||
                   sample((float)sv.v[3], (float)cc.v[1], (float)cc.v[2], (float)cc.v[3], make_float2(db.z, db.w));
        }
        while (!stop && step < n_step)  // (n_step % 4 samples left)
            stop = sample((float)s[step], (float)c[step * 3], (float)c[step * 3 + 1], (float)c[step * 3 + 2], dl[step]);
    } else {
        while (step < n_step) {
            if (sample((float)s[step], (float)c[step * 3], (float)c[step * 3 + 1], (float)c[step * 3 + 2], dl[step])) break;
        }
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = weight_sum; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// ---------------------------------------------------------------- alive-ray compaction
// Stable compaction with wave ballots.  Pass 1: per-wave survivor counts.  Pass 2: wave base by
// strided reduction over earlier waves (same scheme as march_write), rank by mbcnt of the ballot.
__global__ void __launch_bounds__(64) k_compact_count(const int32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ ws,
                                                      const int32_t* __restrict__ n_in_dev) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const bool keep = i < alive_count(n, n_in_dev) && in[i] >= 0;
    const unsigned long long m = __ballot(keep);
    if (threadIdx.x == 0) ws[blockIdx.x] = (uint32_t)__popcll(m);
}
__global__ void __launch_bounds__(64) k_compact_write(const int32_t* __restrict__ in, uint32_t n, int32_t* __restrict__ out,
                                                      int32_t* __restrict__ n_out, const uint32_t* __restrict__ ws,
                                                      const int32_t* __restrict__ n_in_dev) {
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 64) part += ws[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const int32_t v = i < alive_count(n, n_in_dev) ? in[i] : -1;  // (n_in_dev may alias n_out: read before the store below)
    const bool keep = v >= 0;
    const unsigned long long m = __ballot(keep);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (keep) out[part + rank] = v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = (int32_t)(part + (uint32_t)__popcll(m));
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT int s3d_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                      float min_near, float* nears, float* fars, float* noises, const int32_t* noise_step,
                                      uint32_t noise_key, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null pointer");
    hipLaunchKernelGGL(k_near_far, dim3(stream_grid(N, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d, aabb, N,
                       min_near, nears, fars, noises, noise_step, noise_key);
    return check_launch("near_far_from_aabb");
}

S3D_EXPORT int s3d_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                                s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(rays_o && rays_d && coords, "sph_from_ray: null pointer");
    hipLaunchKernelGGL(k_sph_from_ray, dim3(stream_grid(N, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d,
                       radius, N, coords);
    return check_launch("sph_from_ray");
}

S3D_EXPORT int s3d_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(coords && indices, "morton3D: null pointer");
    hipLaunchKernelGGL(k_morton3d, dim3(stream_grid(N, 256)), dim3(256), 0, as_stream(stream), coords, N, indices);
    return check_launch("morton3D");
}

S3D_EXPORT int s3d_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(coords && indices, "morton3D_invert: null pointer");
    hipLaunchKernelGGL(k_morton3d_invert, dim3(stream_grid(N, 256)), dim3(256), 0, as_stream(stream), indices, N, coords);
    return check_launch("morton3D_invert");
}

S3D_EXPORT int s3d_mip_levels(const float* xyz, const float* dt, uint32_t N, uint32_t H, uint32_t C, int32_t* mip_pos,
                              int32_t* mip_dt, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE((!mip_pos || xyz) && (!mip_dt || dt), "mip_levels: an output without its input");
    hipLaunchKernelGGL(k_mip_levels, dim3(stream_grid(N, 256)), dim3(256), 0, as_stream(stream), xyz, dt, N, (float)H, (float)C,
                       mip_pos, mip_dt);
    return check_launch("mip_levels");
}

S3D_EXPORT int s3d_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield,
                            s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(grid && bitfield, "packbits: null pointer");
    S3D_REQUIRE(((uintptr_t)grid & 15) == 0, "packbits: grid must be 16-byte aligned");
    hipLaunchKernelGGL(k_packbits, dim3(stream_grid(N, 256)), dim3(256), 0, as_stream(stream), grid, N, density_thresh,
                       bitfield);
    return check_launch("packbits");
}

static size_t march_wave_ws(uint32_t N, uint32_t max_steps) { return sizeof(uint32_t) * (kWsHeader + (size_t)N + (size_t)N * max_steps); }

S3D_EXPORT size_t s3d_march_rays_train_workspace_size(uint32_t N, uint32_t max_steps) {
    const size_t lane_path = sizeof(uint32_t) * (kWsHeader + (size_t)div_up<uint32_t>(N ? N : 1, 64));
    if (N <= kWaveMarchMaxRays) return march_wave_ws(N, max_steps) > lane_path ? march_wave_ws(N, max_steps) : lane_path;
    return lane_path;
}

S3D_EXPORT int s3d_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                    float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                    const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                    int32_t* rays, int32_t* counter, const float* noises, void* workspace,
                                    size_t workspace_bytes, int path, const float* aabb, float min_near,
                                    const int32_t* noise_step, uint32_t noise_key, s3d_stream_t stream) {
    // path: 0 = auto (wave-per-ray up to 16,384 rays), 1 = lane-per-ray kernels, 2 = wave-per-ray kernels
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(path >= 0 && path <= 3, "march_rays_train: path must be 0 (auto), 1 (lane per ray), 2 (wave per ray) or 3 (wave per ray, general kernel)");
    S3D_REQUIRE(rays_o && rays_d && grid && nears && fars && rays && counter && noises, "march_rays_train: null pointer");
    S3D_REQUIRE(M == 0 || (xyzs && dirs && deltas), "march_rays_train: null output");
    S3D_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024, "march_rays_train: unsupported cascade/grid size C=%u H=%u", C, H);
    const size_t lane_ws = sizeof(uint32_t) * (kWsHeader + (size_t)div_up<uint32_t>(N, 64));
    S3D_REQUIRE(workspace && workspace_bytes >= lane_ws, "march_rays_train: workspace too small (%zu < %zu)", workspace_bytes, lane_ws);
    const uint32_t nw = div_up<uint32_t>(N, 64);
    uint32_t* ws = reinterpret_cast<uint32_t*>(workspace);
    const bool wave_ok = workspace_bytes >= march_wave_ws(N, max_steps) && max_steps >= 1;
    const bool use_wave = ((path == 2 || path == 3) && wave_ok) || (path == 0 && wave_ok && N <= kWaveMarchMaxRays);
    // aabb given: nears / fars (and, with noise_step, noises) are OUTPUTS of this call — made by the wave-per-ray count kernel
    // itself, by k_near_far in front of the lane-per-ray kernels
    float* nears_w = const_cast<float*>(nears);
    float* fars_w = const_cast<float*>(fars);
    float* noises_w = const_cast<float*>(noises);
    if (aabb && !use_wave)
        hipLaunchKernelGGL(k_near_far, dim3(stream_grid(N, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d, aabb, N,
                           min_near, nears_w, fars_w, noise_step ? noises_w : nullptr, noise_step, noise_key);
    if (use_wave) {
        const bool fast = C == 1 && H <= 128 && path != 3;  // one cascade, morton table of 128 spread coordinates (path 3: the general kernel, tests)
#define S3D_COUNT_WAVE(CD_, F_) hipLaunchKernelGGL((k_march_count_wave<CD_, F_>), dim3(N), dim3(64), 0, as_stream(stream), rays_o, rays_d, \
                                                   grid, bound, dt_gamma, max_steps, N, C, H, nears_w, fars_w, noises_w, rays,            \
                                                   (const int32_t*)counter, ws, aabb, min_near, noise_step, noise_key)
        if (dt_gamma == 0.0f) { if (fast) S3D_COUNT_WAVE(true, true); else S3D_COUNT_WAVE(true, false); }
        else { if (fast) S3D_COUNT_WAVE(false, true); else S3D_COUNT_WAVE(false, false); }
#undef S3D_COUNT_WAVE
        hipLaunchKernelGGL(k_march_write_wave, dim3(N), dim3(64), 0, as_stream(stream), rays_o, rays_d, bound, dt_gamma,
                           max_steps, N, C, H, M, nears, noises, xyzs, dirs, deltas, rays, counter, (const uint32_t*)ws);
        return check_launch("march_rays_train");
    }
    hipLaunchKernelGGL(k_march_count, dim3(nw), dim3(64), 0, as_stream(stream), rays_o, rays_d, grid, bound, dt_gamma,
                       max_steps, N, C, H, nears, fars, noises, rays, (const int32_t*)counter, ws);
    hipLaunchKernelGGL(k_march_write, dim3(nw), dim3(64), 0, as_stream(stream), rays_o, rays_d, grid, bound, dt_gamma,
                       max_steps, N, C, H, M, nears, fars, noises, xyzs, dirs, deltas, rays, counter,
                       (const uint32_t*)ws);
    return check_launch("march_rays_train");
}

S3D_EXPORT int s3d_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                                const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                                float* weights_sum, float* depth, float* image, int path,
                                                s3d_stream_t stream) {
    // path: 0 = wave-per-ray (default), 1 = lane-per-ray (serial chain, the oracle's summation order)
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(rays && weights_sum && depth && image, "composite_rays_train_forward: null pointer");
    S3D_REQUIRE(M == 0 || (sigmas && rgbs && deltas), "composite_rays_train_forward: null input");
    if (path == 1)
        hipLaunchKernelGGL(k_composite_train_fwd, dim3(div_up<uint32_t>(N, 64)), dim3(64), 0, as_stream(stream), sigmas, rgbs,
                           deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    else
        hipLaunchKernelGGL(k_composite_train_fwd_wave, dim3(div_up<uint32_t>(N, 4)), dim3(256), 0, as_stream(stream), sigmas,
                           rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    return check_launch("composite_rays_train_forward");
}

S3D_EXPORT int s3d_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                                 const float* sigmas, const float* rgbs, const float* deltas,
                                                 const int32_t* rays, const float* weights_sum, const float* image,
                                                 uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas,
                                                 float* grad_rgbs, int path, s3d_stream_t stream) {
    if (N == 0 || M == 0) return S3D_OK;
    S3D_REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image &&
                    grad_sigmas && grad_rgbs, "composite_rays_train_backward: null pointer");
    if (path == 1)
        hipLaunchKernelGGL(k_composite_train_bwd, dim3(div_up<uint32_t>(N, 64)), dim3(64), 0, as_stream(stream),
                           grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                           grad_sigmas, grad_rgbs);
    else
        hipLaunchKernelGGL(k_composite_train_bwd_wave, dim3(div_up<uint32_t>(N, 4)), dim3(256), 0, as_stream(stream),
                           grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                           grad_sigmas, grad_rgbs);
    return check_launch("composite_rays_train_backward");
}

S3D_EXPORT int s3d_composite_rays_train_loss(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                             uint32_t M, uint32_t N, float T_thresh, const float* gt, const float* bg_rgb,
                                             const float* grad_loss, const float* gt_depth, float depth_weight,
                                             float* weights_sum, float* depth, float* image, float* grad_sigmas, float* grad_rgbs,
                                             float* grad_image, float* grad_weights_sum, float* loss, float* workspace,
                                             s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(rays && weights_sum && depth && image && gt && bg_rgb && grad_loss && loss && workspace,
                "composite_rays_train_loss: null pointer");
    S3D_REQUIRE(M == 0 || (sigmas && rgbs && deltas && grad_sigmas && grad_rgbs), "composite_rays_train_loss: null sample buffer");
    S3D_REQUIRE((grad_image == nullptr) == (grad_weights_sum == nullptr), "composite_rays_train_loss: grad_image and grad_weights_sum "
                "come together");
    hipLaunchKernelGGL(k_composite_train_loss_wave, dim3(div_up<uint32_t>(N, 4)), dim3(256), 0, as_stream(stream), sigmas, rgbs, deltas,
                       rays, M, N, T_thresh, gt, bg_rgb[0], bg_rgb[1], bg_rgb[2], grad_loss, gt_depth, depth_weight, weights_sum, depth,
                       image, grad_sigmas, grad_rgbs, grad_image, grad_weights_sum, workspace);
    hipLaunchKernelGGL(k_bg_mse_reduce, dim3(1), dim3(1024), 0, as_stream(stream), (const float*)workspace,
                       gt_depth ? (const float*)(workspace + (size_t)3 * N) : nullptr, N, depth_weight, loss);
    return check_launch("composite_rays_train_loss");
}

S3D_EXPORT int s3d_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                              const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                              uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                              float* xyzs, float* dirs, float* deltas, const float* noises, const int32_t* n_alive_dev,
                              int32_t* n_rows_out, uint32_t rows_total, int zero_unfilled, s3d_stream_t stream) {
    (void)nears;
    if (n_alive == 0 || n_step == 0) return S3D_OK;
    S3D_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas, "march_rays: null pointer");
    S3D_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024, "march_rays: unsupported cascade/grid size C=%u H=%u", C, H);
    S3D_REQUIRE(!zero_unfilled || rows_total >= n_alive * n_step, "march_rays: rows_total (the extent of xyzs / dirs / deltas) is "
                "smaller than n_alive * n_step");
    S3D_REQUIRE((uint64_t)n_alive * n_step < (1ull << 32) && (zero_unfilled || rows_total >= n_alive * n_step || rows_total == 0),
                "march_rays: n_alive * n_step does not fit the sample buffers");
    // ray slots per wave: enough waves to fill the chip (>= 4 per SIMD while the rays last), pools no larger than 256
    uint32_t pool = (n_alive / 4096u) & ~63u;
    pool = pool < 64u ? 64u : (pool > 256u ? 256u : pool);
    const dim3 g1(div_up<uint32_t>(n_alive, 4 * pool)), b(256);
    float2* tl = reinterpret_cast<float2*>(deltas);
    if (n_alive <= kMarchGroupMaxRays) {  // few rays, many steps each: sixteen lanes per ray
        const dim3 gg(div_up<uint32_t>(n_alive, 16));
#define S3D_MARCH_G(C1_, G0_) hipLaunchKernelGGL((k_march_rays_g<C1_, G0_>), gg, b, 0, as_stream(stream), n_alive, n_step, rays_alive, rays_t, \
                                                rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, tl, noises, n_alive_dev, n_rows_out)
        if (C == 1 && dt_gamma == 0.0f) S3D_MARCH_G(true, true);
        else if (C == 1) S3D_MARCH_G(true, false);
        else if (dt_gamma == 0.0f) S3D_MARCH_G(false, true);
        else S3D_MARCH_G(false, false);
#undef S3D_MARCH_G
    } else if (C == 1)
        hipLaunchKernelGGL(k_march_rays_t<true>, g1, b, 0, as_stream(stream), n_alive, n_step, rays_alive, rays_t, rays_o, rays_d,
                           bound, dt_gamma, max_steps, C, H, grid, fars, tl, noises, n_alive_dev, n_rows_out, pool);
    else
        hipLaunchKernelGGL(k_march_rays_t<false>, g1, b, 0, as_stream(stream), n_alive, n_step, rays_alive, rays_t, rays_o, rays_d,
                           bound, dt_gamma, max_steps, C, H, grid, fars, tl, noises, n_alive_dev, n_rows_out, pool);
    const uint32_t rows = zero_unfilled ? std::max(n_alive * n_step, rows_total) : n_alive * n_step;
    hipLaunchKernelGGL(k_march_rays_expand, dim3(div_up<uint32_t>(rows, 256)), b, 0, as_stream(stream), n_alive, n_step, rays_alive,
                       rays_o, rays_d, bound, dt_gamma, max_steps, C, H, xyzs, dirs, tl, n_alive_dev, rows_total,
                       zero_unfilled != 0);
    return check_launch("march_rays");
}

S3D_EXPORT int s3d_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive,
                                  float* rays_t, const void* sigmas, const void* rgbs, const float* deltas,
                                  float* weights_sum, float* depth, float* image, const int32_t* n_alive_dev,
                                  int sigmas_dtype, int rgbs_dtype, s3d_stream_t stream) {
    if (n_alive == 0) return S3D_OK;
    S3D_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image,
                "composite_rays: null pointer");
    S3D_REQUIRE((sigmas_dtype == S3D_F32 || sigmas_dtype == S3D_F16) && (rgbs_dtype == S3D_F32 || rgbs_dtype == S3D_F16),
                "composite_rays: sigmas / rgbs must be f32 or f16");
    const dim3 grid(div_up<uint32_t>(n_alive, 64)), block(64);
    hipStream_t st = as_stream(stream);
    // vector loads: fp32 buffers as they come (4-byte aligned by type); an fp16 buffer needs 8-byte aligned rows
    const bool half_in = sigmas_dtype == S3D_F16 || rgbs_dtype == S3D_F16;
    const bool vec4 = n_step >= 4 && (((uintptr_t)sigmas | (uintptr_t)rgbs | (uintptr_t)deltas) & 3u) == 0u &&
                      (!half_in || ((n_step & 3u) == 0u && (((uintptr_t)sigmas | (uintptr_t)rgbs) & 7u) == 0u));
#define S3D_COMPOSITE(TS, TC)                                                                                              \
    hipLaunchKernelGGL((k_composite_rays<TS, TC>), grid, block, 0, st, n_alive, n_step, T_thresh, rays_alive, rays_t,        \
                       (const TS*)sigmas, (const TC*)rgbs, deltas, weights_sum, depth, image, n_alive_dev, vec4)
    if (sigmas_dtype == S3D_F32 && rgbs_dtype == S3D_F32) S3D_COMPOSITE(float, float);
    else if (sigmas_dtype == S3D_F32) S3D_COMPOSITE(float, __half);
    else if (rgbs_dtype == S3D_F32) S3D_COMPOSITE(__half, float);
    else S3D_COMPOSITE(__half, __half);
#undef S3D_COMPOSITE
    return check_launch("composite_rays");
}

S3D_EXPORT size_t s3d_compact_alive_workspace_size(uint32_t n) {
    return sizeof(uint32_t) * (size_t)div_up<uint32_t>(n ? n : 1, 64);
}

S3D_EXPORT int s3d_compact_alive(const int32_t* in, uint32_t n, int32_t* out, int32_t* n_out, void* workspace,
                                 size_t workspace_bytes, const int32_t* n_in_dev, s3d_stream_t stream) {
    S3D_REQUIRE(n_out, "compact_alive: null n_out");
    if (n == 0) { S3D_HIP(hipMemsetAsync(n_out, 0, sizeof(int32_t), as_stream(stream))); return S3D_OK; }
    S3D_REQUIRE(in && out && workspace && workspace_bytes >= s3d_compact_alive_workspace_size(n),
                "compact_alive: bad arguments");
    const uint32_t nw = div_up<uint32_t>(n, 64);
    hipLaunchKernelGGL(k_compact_count, dim3(nw), dim3(64), 0, as_stream(stream), in, n, (uint32_t*)workspace, n_in_dev);
    hipLaunchKernelGGL(k_compact_write, dim3(nw), dim3(64), 0, as_stream(stream), in, n, out, n_out,
                       (const uint32_t*)workspace, n_in_dev);
    return check_launch("compact_alive");
}

constexpr uint32_t kSweepBlocks = 1024;

S3D_EXPORT int s3d_sweep_draw(const double* u_uniform, const double* u_occupied, const int32_t* occ_csum, uint32_t N, uint32_t H,
                              float bound, float half_cell, uint32_t noise_key, const int32_t* noise_step, int32_t* cells,
                              float* xyzs, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(u_uniform && u_occupied && occ_csum && cells && xyzs, "sweep_draw: null pointer");
    S3D_REQUIRE(H >= 2 && H <= 1024 && (uint64_t)2 * N < (1ull << 31), "sweep_draw: unsupported grid size H=%u or N=%u", H, N);
    hipLaunchKernelGGL(k_sweep_draw, dim3(div_up<uint32_t>(2 * N, 256)), dim3(256), 0, as_stream(stream), u_uniform, u_occupied,
                       occ_csum, N, H, bound, half_cell, noise_key, noise_step, cells, xyzs);
    return check_launch("sweep_draw");
}

S3D_EXPORT size_t s3d_sweep_update_workspace_size(uint32_t n_cells) { return ((size_t)n_cells + kSweepBlocks) * sizeof(float); }

S3D_EXPORT int s3d_sweep_update(float* density_grid, uint32_t n_cells, const int32_t* cells, const void* sigma, int sigma_dtype,
                                uint32_t n, float density_scale, float decay, void* workspace, size_t workspace_bytes,
                                float* grid_sum, int32_t* step_counter, s3d_stream_t stream) {
    S3D_REQUIRE(density_grid && n_cells > 0 && (n == 0 || (cells && sigma)) && grid_sum, "sweep_update: null pointer");
    S3D_REQUIRE(sigma_dtype == S3D_F32 || sigma_dtype == S3D_F16, "sweep_update: sigma dtype must be f32 or f16");
    S3D_REQUIRE(workspace && workspace_bytes >= s3d_sweep_update_workspace_size(n_cells), "sweep_update: workspace too small");
    hipStream_t st = as_stream(stream);
    float* tmp = (float*)workspace;
    float* partial = tmp + n_cells;
    hipLaunchKernelGGL(k_sweep_fill, dim3(div_up<uint32_t>(n_cells, 256)), dim3(256), 0, st, tmp, n_cells);
    if (n) {
        if (sigma_dtype == S3D_F16)
            hipLaunchKernelGGL(k_sweep_scatter<_Float16>, dim3(div_up<uint32_t>(n, 256)), dim3(256), 0, st, cells, (const _Float16*)sigma, n,
                               density_scale, tmp);
        else
            hipLaunchKernelGGL(k_sweep_scatter<float>, dim3(div_up<uint32_t>(n, 256)), dim3(256), 0, st, cells, (const float*)sigma, n,
                               density_scale, tmp);
    }
    const uint32_t blocks = std::min<uint32_t>(kSweepBlocks, div_up<uint32_t>(n_cells, 256));
    hipLaunchKernelGGL(k_sweep_update, dim3(blocks), dim3(256), 0, st, density_grid, (const float*)tmp, n_cells, decay, partial);
    hipLaunchKernelGGL(k_sweep_mean, dim3(1), dim3(256), 0, st, (const float*)partial, blocks, 1.0f, grid_sum, step_counter);
    return check_launch("sweep_update");
}
