/*
 * s3o_raymarching.c — CPU ORACLE for the raymarching package.
 *
 * TEST INFRASTRUCTURE ONLY (see s3o_common.h).  Scalar restatement of the ten
 * native entry points of the reference's raymarching extension, one C function
 * per `_backend` function, following raymarching/src/raymarching.cu (cited per
 * function).  PINNED for the integer helpers: __expand_bits / __morton3D /
 * __morton3D_invert and mip_from_pos / mip_from_dt (raymarching.cu:42-81) and the
 * two morton kernels' call lines are evaluated from the reference TEXT by
 * oracle/gen_golden.py `int` (tests/golden/int_kernels.npz) and reproduced here
 * bit for bit (tests/test_int_golden.py); so are kernel_packbits and
 * kernel_near_far_from_aabb.  PINNED TO THE TEXT UNDER A STATED MODEL: both
 * marchers — kernel_march_rays_train (:311-478) and kernel_march_rays (:701-800) —
 * are run statement by statement with C's typing explicit and every float product
 * that feeds an add fused (nvcc's default -fmad=true; oracle/gen_golden.py `march`,
 * tests/golden/march_kernels.npz), and this file reproduces ray table, counter and
 * every sample bit for bit (tests/test_march_golden.py); the three compositing
 * kernels are run in plain float32 (`float`, float_kernels.npz) and reproduced
 * within 1e-4 relative, integers exact (tests/test_float_golden.py).  What stays
 * unobservable without nvcc is the contraction model itself and __expf's rounding.
 * The reference ships no golden vectors or asserting tests for these functions and
 * its CUDA sources cannot be built in this image; besides the text fixtures the
 * restatement is checked against hand-derived known answers
 * (tests/test_oracle_raymarching.py) and against the reference's own Python
 * control flow run on top of it (oracle/gen_golden.py).
 *
 * Span order: the reference reserves output spans with atomicAdd in arrival
 * order (raymarching.cu:405-406), which is non-deterministic.  The oracle
 * emits the ray-ordered packing (what a serial run produces); every valid
 * reference run is a permutation of the spans with identical per-ray content.
 */
#include "s3o_common.h"
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define S3O_SQRT3 1.7320508075688772f /* raymarching.cu:19 */
#define S3O_RPI 0.3183098861837907f   /* raymarching.cu:22 */

S3O_API void s3o_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

S3O_API int s3o_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* raymarching.cu:56-63 */
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:65-71 */
static inline uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* raymarching.cu:73-81 */
static inline uint32_t morton3d_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* raymarching.cu:42-47: frexpf exponent clamped to [0, C-1]. */
static inline int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}
/* raymarching.cu:49-54: `dt * H * 0.5` — 0.5 is a double literal, so the last
 * product is formed in double and rounded to float at the assignment. */
static inline int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* exported so tests can pin the cascade selection against tests/golden/int_kernels.npz */
S3O_API void s3o_mip_from_pos(const float* xyz, uint32_t N, float max_cascade, int32_t* out) {
    for (uint32_t n = 0; n < N; n++) out[n] = mip_from_pos(xyz[n * 3], xyz[n * 3 + 1], xyz[n * 3 + 2], max_cascade);
}
S3O_API void s3o_mip_from_dt(const float* dt, uint32_t N, float H, float max_cascade, int32_t* out) {
    for (uint32_t n = 0; n < N; n++) out[n] = mip_from_dt(dt[n], H, max_cascade);
}

/* ---- near/far: raymarching.cu:92-145 ---- */
S3O_API void s3o_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                                    uint32_t N, float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float* o = rays_o + n * 3;
        const float* d = rays_d + n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
        if (near > far) { float t = near; near = far; far = t; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float t = near_y; near_y = far_y; far_y = t; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = 3.402823466e+38f; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float t = near_z; near_z = far_z; far_z = t; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = 3.402823466e+38f; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* ---- sph_from_ray: raymarching.cu:163-198 ----
 * Floating point: A, B, C are sums of products; we pin the left-to-right
 * fused chain nvcc emits (fma(c, c', fma(b, b', a*a'))). */
S3O_API void s3o_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N,
                              float* coords) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float* o = rays_o + n * 3;
        const float* d = rays_d + n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float dx = d[0], dy = d[1], dz = d[2];
        const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
        const float C = fmaf(-radius, radius, fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
        const float t = (-B + sqrtf(fmaf(B, B, -(A * C)))) / A;
        const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[n * 2 + 0] = fmaf(2 * theta, S3O_RPI, -1.0f);
        coords[n * 2 + 1] = phi * S3O_RPI;
    }
}

/* ---- morton: raymarching.cu:214-254 ---- */
S3O_API void s3o_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++)
        indices[n] = (int32_t)morton3d((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1],
                                       (uint32_t)coords[n * 3 + 2]);
}
S3O_API void s3o_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const int32_t ind = indices[n]; /* arithmetic shift of a signed int, as in the reference */
        coords[n * 3 + 0] = (int32_t)morton3d_invert((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)morton3d_invert((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)morton3d_invert((uint32_t)(ind >> 2));
    }
}

/* ---- packbits: raymarching.cu:268-289; N = number of output bytes ---- */
S3O_API void s3o_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float* g = grid + n * 8;
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (g[i] > density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* One DDA probe shared by the three marching kernels
 * (raymarching.cu:360-399, 428-478, 751-803). */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Cf, Hf;
    uint32_t H;
    const uint8_t* grid;
} march_ctx;

static inline void march_ctx_init(march_ctx* c, const float* o, const float* d, float bound,
                                  float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                  const uint8_t* grid) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1 / c->dx; c->rdy = 1 / c->dy; c->rdz = 1 / c->dz;
    c->rH = 1 / (float)H;
    c->H3 = (float)(H * H * H);               /* raymarching.cu:339: uint32 product -> float */
    c->bound = bound; c->dt_gamma = dt_gamma;
    c->dt_min = 2 * S3O_SQRT3 / (float)max_steps;                 /* :345 */
    c->dt_max = 2 * S3O_SQRT3 * (float)(1 << (C - 1)) / (float)H; /* :346 */
    c->Cf = (float)C; c->Hf = (float)H; c->H = H; c->grid = grid;
}

/* Evaluate the sample position at t, look up occupancy.  Returns occ and fills
 * x,y,z,dt; when not occupied *t_skip is the t at which the voxel is left. */
static inline int march_probe(const march_ctx* c, float t, float* px, float* py, float* pz,
                              float* pdt, float* t_skip) {
    const float x = s3o_clampf(fmaf(t, c->dx, c->ox), -c->bound, c->bound);
    const float y = s3o_clampf(fmaf(t, c->dy, c->oy), -c->bound, c->bound);
    const float z = s3o_clampf(fmaf(t, c->dz, c->oz), -c->bound, c->bound);
    const float dt = s3o_clampf(t * c->dt_gamma, c->dt_min, c->dt_max);
    const int la = mip_from_pos(x, y, z, c->Cf), lb = mip_from_dt(dt, c->Hf, c->Cf);
    const int level = la > lb ? la : lb;
    const float mip_bound = fminf(scalbnf(1.0f, level), c->bound);
    const float mip_rbound = 1 / mip_bound;
    /* `0.5 * (x * mip_rbound + 1) * H`: float fma, then double products, the
     * clamp() call rounds to float, the int conversion truncates. */
    const double Hd = (double)c->H;
    const int nx = (int)s3o_clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * Hd), 0.0f, (float)(c->H - 1));
    const int ny = (int)s3o_clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * Hd), 0.0f, (float)(c->H - 1));
    const int nz = (int)s3o_clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * Hd), 0.0f, (float)(c->H - 1));
    /* `level * H3 + morton` is a float sum converted to uint32 (exact < 2^24). */
    const uint32_t index = (uint32_t)((float)level * c->H3 + (float)morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = (c->grid[index / 8] & (1 << (index % 8))) != 0;
    *px = x; *py = y; *pz = z; *pdt = dt;
    if (!occ) {
        const float sx = copysignf(1.0f, c->dx), sy = copysignf(1.0f, c->dy), sz = copysignf(1.0f, c->dz);
        /* (((n + 0.5 + 0.5*sign) * rH * 2 - 1) * mip_bound - p) * rd, fused where nvcc fuses */
        const float tx = fmaf(fmaf(fmaf(0.5f, sx, (float)nx + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -x) * c->rdx;
        const float ty = fmaf(fmaf(fmaf(0.5f, sy, (float)ny + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -y) * c->rdy;
        const float tz = fmaf(fmaf(fmaf(0.5f, sz, (float)nz + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -z) * c->rdz;
        *t_skip = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return occ;
}

static inline float march_skip(const march_ctx* c, float t, float tt) {
    do { t += s3o_clampf(t * c->dt_gamma, c->dt_min, c->dt_max); } while (t < tt);
    return t;
}

/* ---- march_rays_train: raymarching.cu:312-480 ---- */
S3O_API void s3o_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid,
                                  float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                                  uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                  const float* fars, float* xyzs, float* dirs, float* deltas,
                                  int32_t* rays, int32_t* counter, const float* noises) {
    uint32_t* num = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    float* t0s = (float*)malloc(sizeof(float) * (N ? N : 1));
    /* first pass: count (raymarching.cu:353-400) */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        march_ctx c;
        march_ctx_init(&c, rays_o + n * 3, rays_d + n * 3, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t0 = nears[n];
        t0 = fmaf(s3o_clampf(t0 * dt_gamma, c.dt_min, c.dt_max), noises[n], t0); /* :351 */
        float t = t0;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {
            float x, y, z, dt, tt;
            if (march_probe(&c, t, &x, &y, &z, &dt, &tt)) { num_steps++; t += dt; }
            else t = march_skip(&c, t, tt);
        }
        num[n] = num_steps;
        t0s[n] = t0;
    }
    /* span reservation in ray order (the canonical member of :405-406) */
    uint32_t* offs = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    uint32_t point_index = (uint32_t)counter[0], ray_index = (uint32_t)counter[1];
    const uint32_t ray_base = ray_index;
    for (uint32_t n = 0; n < N; n++) { offs[n] = point_index; point_index += num[n]; ray_index++; }
    counter[0] = (int32_t)point_index;
    counter[1] = (int32_t)ray_index;
    /* second pass: write (raymarching.cu:410-479) */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t num_steps = num[n], off = offs[n];
        int32_t* r = rays + (size_t)(ray_base + n) * 3;
        r[0] = (int32_t)n; r[1] = (int32_t)off; r[2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (off + num_steps > M) continue;
        march_ctx c;
        march_ctx_init(&c, rays_o + n * 3, rays_d + n * 3, bound, dt_gamma, max_steps, C, H, grid);
        float* px = xyzs + (size_t)off * 3;
        float* pd = dirs + (size_t)off * 3;
        float* pl = deltas + (size_t)off * 2;
        const float far = fars[n];
        float t = t0s[n], last_t = t;
        uint32_t step = 0;
        while (t < far && step < num_steps) {
            float x, y, z, dt, tt;
            if (march_probe(&c, t, &x, &y, &z, &dt, &tt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt;
                pl[0] = dt; pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2; step++;
            } else t = march_skip(&c, t, tt);
        }
    }
    free(num); free(t0s); free(offs);
}

/* ---- composite_rays_train_forward: raymarching.cu:501-577 ----
 * `__expf` is a fast-math intrinsic in the reference; the oracle uses expf and
 * the parity tests carry the FP tolerance.  Accumulations are written as the
 * fused forms nvcc emits for `r += weight * rgb`. */
S3O_API void s3o_composite_rays_train_forward(const float* sigmas, const float* rgbs,
                                              const float* deltas, const int32_t* rays, uint32_t M,
                                              uint32_t N, float T_thresh, float* weights_sum,
                                              float* depth, float* image) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                       num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float* s = sigmas + offset;
        const float* c = rgbs + (size_t)offset * 3;
        const float* dl = deltas + (size_t)offset * 2;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, c[0], r); g = fmaf(weight, c[1], g); b = fmaf(weight, c[2], b);
            t += dl[1];
            d = fmaf(weight, t, d);
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            s++; c += 3; dl += 2;
        }
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* ---- composite_rays_train_backward: raymarching.cu:602-682 ---- */
S3O_API void s3o_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                               const float* sigmas, const float* rgbs,
                                               const float* deltas, const int32_t* rays,
                                               const float* weights_sum, const float* image,
                                               uint32_t M, uint32_t N, float T_thresh,
                                               float* grad_sigmas, float* grad_rgbs) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                       num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gws = grad_weights_sum[index];
        const float* gi = grad_image + (size_t)index * 3;
        const float r_final = image[index * 3], g_final = image[index * 3 + 1],
                    b_final = image[index * 3 + 2], ws_final = weights_sum[index];
        const float* s = sigmas + offset;
        const float* c = rgbs + (size_t)offset * 3;
        const float* dl = deltas + (size_t)offset * 2;
        float* gs = grad_sigmas + offset;
        float* gc = grad_rgbs + (size_t)offset * 3;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, c[0], r); g = fmaf(weight, c[1], g); b = fmaf(weight, c[2], b);
            ws += weight;
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
            /* raymarching.cu:662-667, products fused into the running sum left to right */
            float acc = gi[0] * fmaf(T, c[0], -(r_final - r));
            acc = fmaf(gi[1], fmaf(T, c[1], -(g_final - g)), acc);
            acc = fmaf(gi[2], fmaf(T, c[2], -(b_final - b)), acc);
            acc = fmaf(gws, 1 - ws_final, acc);
            gs[0] = dl[0] * acc;
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; gs++; gc += 3;
        }
    }
}

/* ---- march_rays (inference): raymarching.cu:701-805 ---- */
S3O_API void s3o_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
                            const float* rays_t, const float* rays_o, const float* rays_d,
                            float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                            const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                            float* dirs, float* deltas, const float* noises) {
    (void)nears;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int32_t index = rays_alive[n];
        march_ctx c;
        march_ctx_init(&c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma,
                       max_steps, C, H, grid);
        float* px = xyzs + (size_t)n * n_step * 3;
        float* pd = dirs + (size_t)n * n_step * 3;
        float* pl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        const float far = fars[index];
        uint32_t step = 0;
        t = fmaf(s3o_clampf(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t); /* :746 */
        float last_t = t;
        while (t < far && step < n_step) {
            float x, y, z, dt, tt;
            if (march_probe(&c, t, &x, &y, &z, &dt, &tt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt;
                pl[0] = dt; pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2; step++;
            } else t = march_skip(&c, t, tt);
        }
    }
}

/* ---- composite_rays (inference, in place): raymarching.cu:819-905 ---- */
S3O_API void s3o_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                int32_t* rays_alive, float* rays_t, const float* sigmas,
                                const float* rgbs, const float* deltas, float* weights_sum,
                                float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int32_t index = rays_alive[n];
        const float* s = sigmas + (size_t)n * n_step;
        const float* c = rgbs + (size_t)n * n_step * 3;
        const float* dl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        float weight_sum = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t += dl[1];
            d = fmaf(weight, t, d);
            r = fmaf(weight, c[0], r); g = fmaf(weight, c[1], g); b = fmaf(weight, c[2], b);
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; step++;
        }
        if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
        weights_sum[index] = weight_sum; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}
