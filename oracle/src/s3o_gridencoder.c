/*
 * s3o_gridencoder.c — CPU ORACLE for the gridencoder package.
 *
 * TEST INFRASTRUCTURE ONLY (see s3o_common.h).  Scalar restatement of
 * gridencoder/src/gridencoder.cu: grid_encode_forward (:87-242),
 * grid_encode_backward (:245-366) and grad_total_variation (:503-607), D in
 * {2..5}, C in {1,2,4,8}, fp32 and fp16 tables.  PINNED for the integer half:
 * fast_hash / get_grid_index (:50-84) and the cell / corner-row lines of
 * kernel_grid (:137-149, 165-180) are evaluated from the reference TEXT by
 * oracle/gen_golden.py `int` (tests/golden/int_kernels.npz) and reproduced here
 * bit for bit (tests/test_int_golden.py).  The floating-point half is PINNED TO
 * THE TEXT UNDER A STATED MODEL: kernel_grid (:87-242, incl. smoothstep and dy_dx)
 * is run statement by statement with C's typing explicit, exp2f correctly rounded
 * and every float product that feeds an add fused (nvcc's default -fmad=true;
 * oracle/gen_golden.py `grid`, tests/golden/grid_kernels.npz) and the fp32 forward
 * here reproduces it bit for bit on four encoder configurations; the backward
 * kernels (:245-366) agree in the rows touched and within fp32 summation order
 * (tests/test_grid_golden.py).  What stays unobservable without nvcc is the
 * contraction model itself and CUDA's exp2f; the at::Half instantiation (fp16
 * tables, `-O`) is modelled with c10::Half's operator semantics and reproduced
 * bit for bit as well.  Also checked against hand-derived known answers,
 * finite differences with the reference's own gradcheck tolerances
 * (testing/test_hashgrid_grad.py:58) and the reference's Python wrapper run on
 * top of it (oracle/gen_golden.py).
 *
 * Per-level scale: the reference evaluates exp2f(level*S)*H-1 on the device
 * (:138).  A 1-ulp exp2f difference between math libraries can move a sample
 * across a cell boundary, so the contract is that the HOST computes
 * the table once with s3o_grid_level_scales() and both oracle and HIP kernels
 * consume that table.
 *
 * Reference quirk kept on purpose: `float pos_deriv[D] = {1.0f}` (:143) sets
 * only element 0, so with linear interpolation dy_dx is zero for d >= 1.
 */
#include "s3o_common.h"
#include <stdlib.h>

#define MAXD 5
#define MAXC 8

static const uint32_t PRIMES[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                   2097192037u, 1434869437u, 2165219737u}; /* :54 */

/* host-side level table: scale_l = exp2f(l*S)*H - 1, written as the fused form */
S3O_API void s3o_grid_level_scales(uint32_t L, float S, uint32_t H, float* scales) {
    for (uint32_t l = 0; l < L; l++) scales[l] = fmaf(exp2f((float)l * S), (float)H, -1.0f);
}

/* gridencoder.cu:66-84 */
static inline uint32_t grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners,
                                  uint32_t ch, uint32_t hashmap_size, uint32_t resolution,
                                  const uint32_t* pos_grid) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t r = 0;
        for (uint32_t i = 0; i < D; i++) r ^= pos_grid[i] * PRIMES[i];
        index = r;
    }
    return (index % hashmap_size) * C + ch;
}

/* exported so tests can pin the integer arithmetic directly */
S3O_API uint32_t s3o_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners,
                                uint32_t ch, uint32_t hashmap_size, uint32_t resolution,
                                const uint32_t* pos_grid) {
    return grid_index(D, C, gridtype, align_corners, ch, hashmap_size, resolution, pos_grid);
}

static inline float ld(const void* p, int dtype, size_t i) {
    return dtype == S3O_F16 ? s3o_h2f(((const s3o_half*)p)[i]) : ((const float*)p)[i];
}
static inline void st(void* p, int dtype, size_t i, float v) {
    if (dtype == S3O_F16) ((s3o_half*)p)[i] = s3o_f2h(v); else ((float*)p)[i] = v;
}

/* position set-up shared by fwd/bwd (:146-156, :284-293) */
static inline void locate(uint32_t D, const float* x, float scale, int align_corners, uint32_t interp,
                          float* pos, float* pos_deriv, uint32_t* pos_grid) {
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            if (pos_deriv) pos_deriv[d] = 6 * pos[d] * (1.0f - pos[d]);     /* :45-47 */
            pos[d] = pos[d] * pos[d] * fmaf(-2.0f, pos[d], 3.0f);             /* :40-42 */
        }
    }
}

/*
 * inputs [B,D] f32 in [0,1]; embeddings [sO,C]; offsets [L+1]; outputs [L,B,C];
 * dy_dx [B,L,D,C] or NULL; optional corner_idx [B,L,2^D] u32 receives the table
 * row (pre *C) of every corner — test hook for bit-exact index parity.
 */
S3O_API void s3o_grid_encode_forward(const float* inputs, const void* embeddings,
                                     const int32_t* offsets, void* outputs, uint32_t B, uint32_t D,
                                     uint32_t C, uint32_t L, const float* scales, void* dy_dx,
                                     uint32_t gridtype, int align_corners, uint32_t interp,
                                     int dtype, uint32_t* corner_idx) {
    const uint32_t ncorner = 1u << D;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t level = 0; level < (int64_t)L; level++) {
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const size_t goff = (size_t)(uint32_t)offsets[level] * C;
            const float* x = inputs + b * D;
            const size_t ooff = (size_t)level * B * C + (size_t)b * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) st(outputs, dtype, ooff + ch, 0);
                if (dy_dx) {
                    const size_t doff = (size_t)b * D * L * C + (size_t)level * D * C;
                    for (uint32_t i = 0; i < D * C; i++) st(dy_dx, dtype, doff + i, 0);
                }
                if (corner_idx)
                    for (uint32_t i = 0; i < ncorner; i++)
                        corner_idx[((size_t)b * L + level) * ncorner + i] = 0xffffffffu;
                continue;
            }
            const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
            const float scale = scales[level];
            const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
            float pos[MAXD], pos_deriv[MAXD] = {1.0f};
            uint32_t pos_grid[MAXD];
            locate(D, x, scale, align_corners, interp, pos, pos_deriv, pos_grid);

            float res32[MAXC] = {0};
            s3o_half res16[MAXC] = {0};
            for (uint32_t idx = 0; idx < ncorner; idx++) {
                float w = 1;
                uint32_t pgl[MAXD];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                if (corner_idx) corner_idx[((size_t)b * L + level) * ncorner + idx] = index / C;
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float g = ld(embeddings, dtype, goff + index + ch);
                    if (dtype == S3O_F16) res16[ch] = s3o_hadd(res16[ch], s3o_f2h(w * g)); /* half accumulators, :161,184 */
                    else res32[ch] = fmaf(w, g, res32[ch]);
                }
            }
            for (uint32_t ch = 0; ch < C; ch++) {
                if (dtype == S3O_F16) ((s3o_half*)outputs)[ooff + ch] = res16[ch];
                else ((float*)outputs)[ooff + ch] = res32[ch];
            }

            if (dy_dx) { /* :198-241 */
                const size_t doff = (size_t)b * D * L * C + (size_t)level * D * C;
                for (uint32_t gd = 0; gd < D; gd++) {
                    float g32[MAXC] = {0};
                    s3o_half g16[MAXC] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale;
                        uint32_t pgl[MAXD];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                        }
                        pgl[gd] = pos_grid[gd];
                        const uint32_t il = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        pgl[gd] = pos_grid[gd] + 1;
                        const uint32_t ir = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            if (dtype == S3O_F16) {
                                /* w * (Half - Half) * float: the difference is a half op, the rest float */
                                const s3o_half diff = s3o_hsub(((const s3o_half*)embeddings)[goff + ir + ch],
                                                               ((const s3o_half*)embeddings)[goff + il + ch]);
                                g16[ch] = s3o_hadd(g16[ch], s3o_f2h(w * s3o_h2f(diff) * pos_deriv[gd]));
                            } else {
                                const float diff = ((const float*)embeddings)[goff + ir + ch] -
                                                   ((const float*)embeddings)[goff + il + ch];
                                g32[ch] = fmaf(w * diff, pos_deriv[gd], g32[ch]);
                            }
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) {
                        if (dtype == S3O_F16) ((s3o_half*)dy_dx)[doff + gd * C + ch] = g16[ch];
                        else ((float*)dy_dx)[doff + gd * C + ch] = g32[ch];
                    }
                }
            }
        }
    }
}

/*
 * grad [L,B,C]; grad_embeddings [sO,C] (caller zero-initialised, accumulated
 * into); optional dy_dx [B,L,D,C] + grad_inputs [B,D].
 * Scatter order: the reference uses atomics in arbitrary order; the oracle adds
 * in (level, point, corner) order.  fp32 sums are compared with a tolerance;
 * fp16 sums (half2 atomics, :322-328) are order-dependent in the reference
 * itself and are accumulated here in half in that canonical order.
 */
S3O_API void s3o_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                      const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                      uint32_t D, uint32_t C, uint32_t L, const float* scales,
                                      const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                                      int align_corners, uint32_t interp, int dtype) {
    (void)embeddings;
    const uint32_t ncorner = 1u << D;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t level = 0; level < (int64_t)L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = scales[level];
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            float pos[MAXD];
            uint32_t pos_grid[MAXD];
            locate(D, x, scale, align_corners, interp, pos, NULL, pos_grid);
            const size_t gro = (size_t)level * B * C + (size_t)b * C;
            for (uint32_t idx = 0; idx < ncorner; idx++) {
                float w = 1;
                uint32_t pgl[MAXD];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (dtype == S3O_F16) {
                        s3o_half* gg = (s3o_half*)grad_embeddings + goff + index + ch;
                        const s3o_half v = s3o_f2h(w * s3o_h2f(((const s3o_half*)grad)[gro + ch]));
                        *gg = s3o_hadd(*gg, v);
                    } else {
                        float* gg = (float*)grad_embeddings + goff + index + ch;
                        *gg += w * ((const float*)grad)[gro + ch];
                    }
                }
            }
        }
    }
    if (dy_dx && grad_inputs) { /* kernel_input_backward :340-366 */
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; t++) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
            const size_t dbase = (size_t)b * L * D * C;
            if (dtype == S3O_F16) {
                s3o_half r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t ch = 0; ch < C; ch++)
                        r = s3o_hadd(r, s3o_hmul(((const s3o_half*)grad)[(size_t)l * B * C + (size_t)b * C + ch],
                                                 ((const s3o_half*)dy_dx)[dbase + (size_t)l * D * C + d * C + ch]));
                ((s3o_half*)grad_inputs)[t] = r;
            } else {
                float r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t ch = 0; ch < C; ch++)
                        r = fmaf(((const float*)grad)[(size_t)l * B * C + (size_t)b * C + ch],
                                 ((const float*)dy_dx)[dbase + (size_t)l * D * C + d * C + ch], r);
                ((float*)grad_inputs)[t] = r;
            }
        }
    }
}

/*
 * grad_total_variation, gridencoder.cu:503-607 (fp32 only: the wrapper runs it
 * under autocast(enabled=False), grid.py:162).  inputs [B,D] in [0,1]; adds the
 * normalised TV gradient into grad [sO,C].
 */
S3O_API void s3o_grad_total_variation(const float* inputs, const float* embeddings, float* grad,
                                      const int32_t* offsets, float weight, uint32_t B, uint32_t D,
                                      uint32_t C, uint32_t L, const float* scales, uint32_t gridtype,
                                      int align_corners) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t level = 0; level < (int64_t)L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = scales[level];
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        const float* grid = embeddings + goff;
        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            uint32_t pos_grid[MAXD];
            for (uint32_t d = 0; d < D; d++)
                pos_grid[d] = (uint32_t)floorf(fmaf(x[d], scale, align_corners ? 0.0f : 0.5f));
            float results[MAXC] = {0}, idelta[MAXC] = {0};
            const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
            const float w = weight / (float)(2 * D);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur = pos_grid[d];
                if (cur < resolution) {
                    pos_grid[d] = cur + 1;
                    const uint32_t ir = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = grid[index + ch] - grid[ir + ch];
                        results[ch] += gv;
                        idelta[ch] = fmaf(gv, gv, idelta[ch]);
                    }
                }
                if (cur > 0) {
                    pos_grid[d] = cur - 1;
                    const uint32_t il = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = grid[index + ch] - grid[il + ch];
                        results[ch] += gv;
                        idelta[ch] = fmaf(gv, gv, idelta[ch]);
                    }
                }
                pos_grid[d] = cur;
            }
            for (uint32_t ch = 0; ch < C; ch++)
                grad[goff + index + ch] += w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f));
        }
    }
}
