/*
 * seal3d_hip.h — C ABI of libseal3d_hip.so, the MI355X (gfx950) replacement for
 * the five CUDA extension modules on Seal-3D's render/train hot path.
 *
 * One entry point per function the reference binds with pybind11
 * (`_raymarching`, `_gridencoder`, `_shencoder`, `_freqencoder`, `_ffmlp`); the
 * reference declaration each one replaces is cited above it.  Conventions:
 *   - plain device pointers + explicit sizes, no torch types;
 *   - every function takes the HIP stream to launch on and returns 0 on
 *     success, non-zero on error (message: s3d_last_error(), thread-local);
 *   - the CALLER allocates every output (and zero-initialises those the
 *     reference zero-initialises: xyzs/dirs/deltas, grad_embeddings,
 *     grad_sigmas/grad_rgbs, SH grad_inputs) — raymarching.py:205-207,
 *     grid.py:77, raymarching.py:283-284, sphere_harmonics.py:50;
 *   - kernels never allocate; functions that need scratch take a caller-owned
 *     workspace whose size the matching *_workspace_size() reports;
 *   - dtype: S3D_F32 / S3D_F16 select the element type of `void*` tensors;
 *   - n_valid (grid encoder, ffmlp, NGP head; NULL = off): DEVICE pointer to the sample count the ray marcher
 *     left in `counter[0]` (raymarching.cu:431-437).  The training step marches into buffers of a static extent
 *     B (raymarching.py:205-207 sizes them from a running mean) and only the first *n_valid rows hold samples;
 *     with the pointer, rows [round_up(*n_valid, 128), B) are absent for the call — neither read nor written,
 *     their outputs keep whatever the caller's buffer held — so the per-sample work follows the samples without
 *     a host read-back (the step stays HIP-graph capturable with one generous B).  Strides and layouts are
 *     those of B.  Results for the valid rows and every reduced gradient are identical to a call without it on
 *     the same buffers with zero gradient in the tail.
 * INTEGRATION.md shows the pybind / ctypes stub a maintainer of the reference
 * would add on top of this header.
 */
#ifndef SEAL3D_HIP_H
#define SEAL3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* s3d_stream_t; /* hipStream_t */

enum { S3D_F32 = 0, S3D_F16 = 1 };
enum { S3D_OK = 0, S3D_ERR_INVALID = 1, S3D_ERR_HIP = 2, S3D_ERR_UNSUPPORTED = 3 };

const char* s3d_last_error(void);
/* "seal3d-hip <version> gfx950" */
const char* s3d_version(void);

/* ------------------------------------------------------------------ raymarching
 * raymarching/src/raymarching.h:7-18 (bindings.cpp:6-17) */

/* raymarching.h:7  void near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
 * Build extension (all optional, NULL/0 = the reference call): noises [N] receives the per-ray jitter of march_rays_train's
 * `perturb` (raymarching.py:211 draws it with torch.rand) as a counter-based uniform u01(noise_key, *noise_step, ray) —
 * the step number lives in device memory, so a graph-replayed step draws fresh jitter without host-side RNG state. */
int s3d_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                           float min_near, float* nears, float* fars, float* noises, const int32_t* noise_step,
                           uint32_t noise_key, s3d_stream_t stream);

/* raymarching.h:8  void sph_from_ray(rays_o, rays_d, radius, N, coords) */
int s3d_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                     s3d_stream_t stream);

/* Test hook (no reference entry point): the cascade selection of the marching kernels, `mip_from_pos` / `mip_from_dt`
 * (raymarching.cu:42-54), on arrays: xyz [N,3] -> mip_pos [N], dt [N] -> mip_dt [N]; either output may be NULL. */
int s3d_mip_levels(const float* xyz, const float* dt, uint32_t N, uint32_t H, uint32_t C, int32_t* mip_pos, int32_t* mip_dt,
                   s3d_stream_t stream);

/* raymarching.h:9  void morton3D(coords, N, indices) */
int s3d_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, s3d_stream_t stream);

/* raymarching.h:10 void morton3D_invert(indices, N, coords) */
int s3d_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, s3d_stream_t stream);

/* raymarching.h:11 void packbits(grid, N, density_thresh, bitfield); N = output bytes */
int s3d_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield,
                 s3d_stream_t stream);

/* Build extension — the steady-state occupancy sweep of NeRFRenderer.update_extra_state (nerf/renderer.py:497-538: H^3/4
 * uniform cells + H^3/4 uniform picks among the occupied cells per cascade, jittered inside the cell, density queried by the
 * caller, then density_grid = max(density_grid * decay, sample) where both are >= 0) without host syncs or index tensors:
 * s3d_sweep_draw: u_uniform / u_occupied [N] doubles in [0, 1) (ascending order makes the queries walk the Z-curve),
 *   occ_csum [H^3] = inclusive prefix count of cells with density > 0 -> cells [2N] (morton index = density_grid index)
 *   and xyzs [2N, 3] (jitter = counter-based u01(noise_key, *noise_step, 3 i + d); noise_step may be NULL = 0).
 * s3d_sweep_update: one cascade's density_grid [n_cells] updated in place from (cells, sigma * density_scale) [n]; of the
 *   samples that fall in one cell the largest is kept (the reference's indexed assignment keeps an arbitrary one);
 *   *grid_sum = sum(max(density_grid, 0)) afterwards (fixed summation order); step_counter (optional) += 1. */
int s3d_sweep_draw(const double* u_uniform, const double* u_occupied, const int32_t* occ_csum, uint32_t N, uint32_t H,
                   float bound, float half_cell, uint32_t noise_key, const int32_t* noise_step, int32_t* cells,
                   float* xyzs, s3d_stream_t stream);
size_t s3d_sweep_update_workspace_size(uint32_t n_cells);
int s3d_sweep_update(float* density_grid, uint32_t n_cells, const int32_t* cells, const void* sigma, int sigma_dtype,
                     uint32_t n, float density_scale, float decay, void* workspace, size_t workspace_bytes,
                     float* grid_sum, int32_t* step_counter, s3d_stream_t stream);

/* raymarching.h:13 void march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M,
 *                                        nears, fars, xyzs, dirs, deltas, rays, counter, noises)
 * Spans are packed in RAY ORDER (deterministic; one valid outcome of the reference's atomic
 * reservation, raymarching.cu:405-406).  counter[0] += total samples, counter[1] += N.
 * path (kernel choice, same results): 0 = auto (wave-per-ray up to 16,384 rays), 1 = lane-per-ray, 2 = wave-per-ray,
 *   3 = wave-per-ray without the single-cascade fast path (its morton table, batched probes and voxel-run shortcut).
 * Rows no ray fills are written as zeros where a consumer bounded by the device-side count (`n_valid`: the count rounded
 * up to 128 rows, at most M) still reads them: [total, round_up(total, 128)) and, for the one ray that straddles the
 * budget (offset < M < offset + steps, dropped like in the reference), [offset, M).  The reference zero-fills all M rows
 * beforehand (raymarching.py:205-207); callers that read every row must still do so.  s3d_composite_rays_train_backward
 * does the same for grad_sigmas / grad_rgbs (plus the samples behind a ray's early termination).
 * aabb != NULL (build extension): s3d_near_far_from_aabb(aabb, min_near, noise_step, noise_key) is part of this call —
 * nears / fars and, with noise_step, noises are then OUTPUTS (written before they are used; same values). */
size_t s3d_march_rays_train_workspace_size(uint32_t N, uint32_t max_steps);
int s3d_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                         float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                         uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                         float* deltas, int32_t* rays, int32_t* counter, const float* noises,
                         void* workspace, size_t workspace_bytes, int path, const float* aabb, float min_near,
                         const int32_t* noise_step, uint32_t noise_key, s3d_stream_t stream);

/* raymarching.h:14 void composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh,
 *                                                    weights_sum, depth, image)
 * path (both directions): 0 = wave-per-ray compositing, 1 = lane-per-ray (serial chain, the reference's order). */
int s3d_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                     const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                     float* weights_sum, float* depth, float* image, int path,
                                     s3d_stream_t stream);

/* raymarching.h:15 void composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas,
 *                       rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs) */
int s3d_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                      const float* sigmas, const float* rgbs, const float* deltas,
                                      const int32_t* rays, const float* weights_sum, const float* image,
                                      uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas,
                                      float* grad_rgbs, int path, s3d_stream_t stream);

/* Build extension — s3d_composite_rays_train_forward, s3d_bg_mse_forward(grad_loss) and s3d_composite_rays_train_backward of
 * one training ray batch as ONE launch (raymarching.py:238-291 + nerf/renderer.py:316 + nerf/utils.py:484-489): the wave that
 * composites a ray forms that ray's loss gradient (k = *grad_loss * 2 / (3N), the announced upstream gradient of the loss) and
 * walks its samples again for grad_sigmas / grad_rgbs; the loss value is summed by a one-workgroup launch behind it in
 * s3d_bg_mse_forward's order (two launches instead of three, the second off the backward's critical data).  Every output equals
 * the three-call sequence (wave-per-ray paths) bit for bit.  gt [N,3], bg_rgb = 3 HOST floats, gt_depth [N] or NULL (value-only
 * depth term), grad_image [N,3] / grad_weights_sum [N]: optional outputs (both or neither), workspace: 4N floats (scratch). */
int s3d_composite_rays_train_loss(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                  uint32_t M, uint32_t N, float T_thresh, const float* gt, const float* bg_rgb,
                                  const float* grad_loss, const float* gt_depth, float depth_weight,
                                  float* weights_sum, float* depth, float* image, float* grad_sigmas, float* grad_rgbs,
                                  float* grad_image, float* grad_weights_sum, float* loss, float* workspace,
                                  s3d_stream_t stream);

/* raymarching.h:17 void march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma,
 *                       max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises)
 * noises may be NULL (= all zero: no perturbation).  zero_unfilled (build extension): the reference's wrapper zero-fills
 * xyzs / dirs / deltas before every call (raymarching.py:324-326: unfilled slots must read as zeros, deltas == 0 ends a
 * ray's chunk); with zero_unfilled != 0 the kernel writes those zeros itself — the slots a ray does not fill and the rows
 * behind the last ray up to rows_total (up to the next multiple of 128 of the live rows when n_alive_dev is given) — and
 * the caller passes uninitialised buffers of rows_total rows.
 * Two launches on `stream` (a walk that records (t, previous t) per emitted sample in the sample's own `deltas` row — one
 * lane per ray with lane refill, or sixteen lanes per ray once few rays are alive — then one lane per row expanding it to
 * xyz / dir / (dt, t' - previous t)); n_alive * n_step must fit 32 bits. */
int s3d_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma,
                   uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                   const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises,
                   const int32_t* n_alive_dev, int32_t* n_rows_out, uint32_t rows_total, int zero_unfilled,
                   s3d_stream_t stream);

/* raymarching.h:18 void composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs,
 *                       deltas, weights_sum, depth, image) — in place */
int s3d_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive,
                       float* rays_t, const void* sigmas, const void* rgbs, const float* deltas,
                       float* weights_sum, float* depth, float* image, const int32_t* n_alive_dev,
                       int sigmas_dtype /* S3D_F32 | S3D_F16: the network's outputs as they come */, int rgbs_dtype,
                       s3d_stream_t stream);

/* Device-side replacement of the host compaction `rays_alive = rays_alive[rays_alive >= 0]`
 * (nerf/renderer.py:363): stable wave-ballot compaction of `in[0..n)` into `out`, count -> *n_out
 * (device int32).  Not in the reference's native surface; used by the build's renderer. */
size_t s3d_compact_alive_workspace_size(uint32_t n);
int s3d_compact_alive(const int32_t* in, uint32_t n, int32_t* out, int32_t* n_out, void* workspace,
                      size_t workspace_bytes, const int32_t* n_in_dev, s3d_stream_t stream);
/* Sync-free inference loop (nerf/renderer.py:341-367 reads the alive count back on every iteration): `n_alive` / `n` of
 * march_rays, composite_rays and compact_alive are then the host's UPPER BOUND (launch geometry, buffer extents) and the
 * optional `n_alive_dev` / `n_in_dev` point at the real count in device memory (NULL: the bound is the count); entries
 * past it are ignored.  march_rays leaves `*n_alive_dev * n_step` in `n_rows_out` (optional) — the `n_valid` of the
 * network kernels that follow.  The host refreshes its bound every few iterations only. */

/* ------------------------------------------------------------------ gridencoder
 * gridencoder/src/gridencoder.h:12-15 (bindings.cpp:6-8).
 * `S` and `H` as in the reference (S = log2(per_level_scale)); the per-level scale table
 * exp2f(l*S)*H-1 (gridencoder.cu:138) is evaluated ONCE ON THE HOST inside the call so that all
 * implementations agree on cell boundaries bit-for-bit; s3d_grid_level_scales exposes it. */
void s3d_grid_level_scales(uint32_t L, float S, uint32_t H, float* scales_out /* host, [L] */);

/* gridencoder.h:12 void grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H,
 *                        dy_dx, gridtype, align_corners, interp)
 * inputs [B,D] f32 in [0,1]; embeddings [sO,C] dtype; offsets [L+1] i32; outputs [L,B,C] dtype;
 * dy_dx [B,L,D,C] dtype or NULL.
 * bound: 0 = inputs already in [0,1] (the reference's native contract); > 0 = raw coordinates in [-bound, bound],
 * normalised in the kernel exactly like GridEncoder.forward does in torch (grid.py:146), dy_dx must be NULL.
 * live (optional, inference): device fp32, row b is live iff live[b * live_stride] != 0; other rows are written as zeros
 * without touching the table.  The inference loop passes `deltas` (nerf/renderer.py:355-360: march_rays leaves the
 * unused slots of a ray's chunk zero-filled and composite_rays stops at the first deltas == 0, raymarching.cu:740). */
int s3d_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                            void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                            uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners,
                            uint32_t interp, int dtype, float bound, const int32_t* n_valid,
                            const float* live, uint32_t live_stride, s3d_stream_t stream);
/* Build extension: TWO tables of identical geometry (offsets, C, L, S, H, gridtype, ...) encoded for the same points in one
 * launch — the density and the colour encoder of the network Seal-3D trains (nerf/network.py:99-128 calls them on the same x):
 * outputs_a / outputs_b [L, B, C] as s3d_grid_encode_forward writes them; no input Jacobian.  Same values as two calls. */
int s3d_grid_encode_forward_pair(const float* inputs, const void* embeddings_a, const void* embeddings_b, const int32_t* offsets,
                                 void* outputs_a, void* outputs_b, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                 uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float bound,
                                 const int32_t* n_valid, const float* live, uint32_t live_stride, s3d_stream_t stream);

/* Test hook: table row of every corner, corner_idx [B,L,2^D] u32 (0xffffffff for out-of-range points). */
int s3d_grid_corner_indices(const float* inputs, const int32_t* offsets, uint32_t* corner_idx, uint32_t B,
                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                            int align_corners, s3d_stream_t stream);

/* gridencoder.h:13 void grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C,
 *                        L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp)
 * grad [L,B,C]; grad_embeddings [table_rows,C] zero-initialised by the caller, accumulated into.
 * max_level_rows = max over levels of offsets[l+1]-offsets[l] (the module builds `offsets` on the host,
 * grid.py:104-112, so it knows it; 0 = unknown).
 * workspace (s3d_grid_encode_backward_workspace_size bytes; 0 = configuration not supported) enables the
 * binned path used for B >= 8192: contributions are partitioned by table slice and summed in LDS as 64-bit
 * fixed point — deterministic, no global atomics.  Without it (NULL) direct atomics are used.
 * path: 0 = auto, 1 = direct atomics, 2 = binned (an error when the workspace is missing).
 * found_inf (optional, build extension): a device float raised to 1 when grad_embeddings holds a non-finite value after
 * the call (torch.amp.GradScaler's check, nerf/utils.py:495-537, made where the gradient is produced: the binned fp16 path
 * reports while it writes the sums, the other paths scan the table once).  Never cleared here.
 * control (optional, build extension): s3d_grid_encode_backward_control_size() bytes of device memory (a few hundred KB,
 * independent of B) that the CALLER zero-fills once after allocating it; every call finds it all-zero and leaves it all-zero
 * (the accumulate kernel clears the bucket cursors it has consumed), so the binned path needs no clearing launch of its
 * own.  Not to be shared by calls that may run concurrently.  NULL: the words live in `workspace` and are cleared per call. */
size_t s3d_grid_encode_backward_workspace_size(uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                               uint32_t max_level_rows, int dtype);
size_t s3d_grid_encode_backward_control_size(uint32_t D, uint32_t C, uint32_t L, uint32_t max_level_rows, int dtype);
int s3d_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                             const int32_t* offsets, void* grad_embeddings, uint32_t max_level_rows,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                             uint32_t interp, int dtype, void* workspace, size_t workspace_bytes,
                             float bound, const int32_t* n_valid, int path, float* found_inf,
                             void* control, size_t control_bytes, s3d_stream_t stream);

/* Build extension: the backward of a table WITH its parameter update (single replica, fp16 C = 2 tables).  The binned path's
 * accumulate kernel holds every row's gradient sum in registers at its write-out; given the optimizer's state it applies
 * torch.optim.Adam's update (betas, eps as passed; the reference: main_SealNeRF.py:283-288) there — fp32 master row, both
 * moments and the fp16 copy the next forward reads — for EVERY row of the table (rows without records take g = 0), and the
 * gradient table is neither written nor read: the separate update's 30 B per parameter become 26 B, overlapped with the
 * records' streaming.  The row's gradient is the exact sum rounded to binary16 (what the unfused path stores), so both routes
 * update to the same bits; where that rounding would overflow the fp32 sum is used.  GradScaler: `found_inf` must hold the
 * step's decision so far when the call is issued (every other gradient producer of the step has run: the table's backward is
 * the last node of the graph); a non-finite dL/dy seen by this call's scatter raises it and skips the update as a whole.
 * step / grad_scale / lr_scale: device floats (step count BEFORE this update; loss scale or NULL; schedule factor or NULL).
 * *applied (host) = 1 when the update was made here; 0 when the call fell back to the plain backward (direct-atomics sizes,
 * more than one level pass, other dtypes): grad_embeddings then holds the gradient and the caller's optimizer applies it. */
typedef struct s3d_grid_adam {
    float* param;
    float* exp_avg;
    float* exp_avg_sq;
    void* param_half; /* fp16 [rows, C] or NULL */
    float lr, beta1, beta2, eps;
    const float* step;
    const float* grad_scale;
    const float* lr_scale;
} s3d_grid_adam;
int s3d_grid_encode_backward_adam(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                  void* grad_embeddings, uint32_t max_level_rows, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                  float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                  void* workspace, size_t workspace_bytes, float bound, const int32_t* n_valid, float* found_inf,
                                  void* control, size_t control_bytes, const s3d_grid_adam* adam, int* applied, s3d_stream_t stream);

/* gridencoder.h:15 void grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H,
 *                        gridtype, align_corners) — fp32 only (grid.py:162 disables autocast) */
int s3d_grad_total_variation(const float* inputs, const float* embeddings, float* grad,
                             const int32_t* offsets, float weight, uint32_t B, uint32_t D, uint32_t C,
                             uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                             s3d_stream_t stream);

/* ------------------------------------------------------------------ shencoder
 * shencoder/src/shencoder.h:9-10.  fp32 (the wrapper casts inputs to f32, sphere_harmonics.py:16). */
int s3d_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree,
                          float* dy_dx /* [B,3,deg^2] or NULL */, s3d_stream_t stream);
int s3d_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree,
                           const float* dy_dx, float* grad_inputs, s3d_stream_t stream);

/* ------------------------------------------------------------------ freqencoder
 * freqencoder/src/freqencoder.h:7,10.  C = D + 2*D*deg. */
int s3d_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                            float* outputs, s3d_stream_t stream);
int s3d_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float* grad_inputs, s3d_stream_t stream);
/* Build extension — two frequency encodings side by side in one fp16 row: out[b] = [freq(a[b]) (D1 + 2 D1 deg1 columns) |
 * freq(d[b]) (D2 + 2 D2 deg2) | zeros up to ld] — TensoRF's colour MLP input cat([encoder(feat), encoder_dir(d)]) as the fp16
 * autocast Linear sees it (tensoRF/network.py:48-51, 160-166): s3d_freq_encode_forward's fp32 values of float(a) and d, rounded
 * to binary16 once.  a: fp16 [B, D1], d: fp32 [B, D2], out: fp16 [B, ld], ld even.  _backward: the gradient w.r.t. a from the
 * row's fp16 gradient (s3d_freq_encode_backward's expression, sin / cos re-computed), fp16 [B, ldg] with the columns behind D1
 * zero (ldg = 32: the row layout s3d_vm_color_backward reads); d takes no gradient (view directions). */
int s3d_freq_encode_pack_forward(const uint16_t* a, const float* d, uint32_t B, uint32_t D1, uint32_t deg1, uint32_t D2,
                                 uint32_t deg2, uint32_t ld, uint16_t* out, const int32_t* n_valid, s3d_stream_t stream);
int s3d_freq_encode_pack_backward(const uint16_t* grad, const uint16_t* a, uint32_t B, uint32_t D1, uint32_t deg1, uint32_t ld,
                                  uint32_t ldg, uint16_t* grad_a, const int32_t* n_valid, s3d_stream_t stream);

/* ------------------------------------------------------------------ ffmlp
 * ffmlp/src/ffmlp.h:8-14.  All tensors fp16 (uint16_t bit patterns).
 * weights: [W,in] | (n-1) x [W,W] | [out,W], row-major [out,in] per layer (ffmlp.cu:631-634).
 * forward_buffer / backward_buffer: [n, B, W] post-activation / pre-activation-gradient scratch.
 * B must be a multiple of 128 (ffmlp.py:156-159 pads); in % 16 == 0; out == 16; W in {16,32,64,128,256} (ffmlp.cu:40-44).
 * W in {32,64} with in <= 64 (every network of the BASELINE configs) run on the register-resident MFMA kernels and have all
 * the extensions below; the other shapes take a layer-by-layer path (hand-written MFMA kernels k_layer / k_wgrad_tile / k_wgrad_finish, fp16
 * storage / fp32 accumulation, csrc/ffmlp_generic.hip; the library links no BLAS) that implements the reference's interface only: forward_buffer / backward_buffer
 * are then REQUIRED for training (plain row-major [n, B, W]) and inference_buffer must hold [2, B, W].
 * input_layout: 0 = inputs [B,in] row-major (the reference); 1 = level-major [in/2][B][2], i.e. the grid
 * encoder's own output layout read in place (and grad_inputs written in it) — no permute copies in between.
 * rgb_head (optional, build extension): fp32 [B, 3] = sigmoid(output[:, 0:3]) written INSTEAD of `outputs` (which may then
 * be NULL) — the colour network's `torch.sigmoid(h)` (nerf/network_ff.py:103) folded into the last layer's store, with the
 * fp16 roundings of the op sequence.  s3d_ffmlp_backward then takes grad_rgb (fp32 [B, 3], gradient w.r.t. that output)
 * and rgb_head instead of `grad`.
 * mid_* (optional, build extension; mid_color_in != NULL switches it on): the density network's head of
 * nerf/network_ff.py:55-96 folded into the last layer — mid_sigma [B] f32 = trunc_exp(output[:, 0]), mid_color_in [B, 32] f16
 * = [half(SH_4(mid_dirs)) | output[:, 1:16] | 0] (the colour network's input), mid_h0 [B] f16 = output[:, 0]; `outputs` may
 * then be NULL.  s3d_ffmlp_backward takes mid_grad_sigma [B] f32 (may be NULL), mid_grad_color_in [B, 32] f16 and mid_h0
 * instead of `grad`.  Same arithmetic and roundings as s3d_ngp_mid_forward / _backward. */
int s3d_ffmlp_forward(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                      uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                      uint32_t output_activation, uint16_t* forward_buffer, uint16_t* outputs,
                      int input_layout, const int32_t* n_valid, float* rgb_head, const float* mid_dirs,
                      float* mid_sigma, uint16_t* mid_color_in, uint16_t* mid_h0, s3d_stream_t stream);
/* ffmlp.h:9: same network without storing intermediates (inference_buffer: unused by the MFMA kernels, [2, B, W] ping-pong
 * scratch for the layer-by-layer shapes) */
int s3d_ffmlp_inference(const uint16_t* inputs, const uint16_t* weights, uint32_t B, uint32_t input_dim,
                        uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                        uint32_t output_activation, uint16_t* inference_buffer, uint16_t* outputs,
                        int input_layout, const int32_t* n_valid, float* rgb_head, const float* mid_dirs,
                      float* mid_sigma, uint16_t* mid_color_in, uint16_t* mid_h0, s3d_stream_t stream);
/* Build extension for the inference loop (nerf/renderer.py:341-367 calls density and colour network back to back on the same
 * rows; nerf/network_ff.py:55-105): both fused MLPs and the two heads in ONE launch — inputs [B, 32] fp16 (or level-major
 * [16][B][2]), dirs [B, 3] fp32 -> sigma [B] fp32 = trunc_exp(h[:, 0]), rgb [B, 3] fp32 = sigmoid(colour(...)[:, :3]);
 * color_in (optional) [B, 32] fp16 receives the colour-net input rows and h0 (optional) [B] fp16 the pre-activation of
 * sigma — what the two networks' backward calls need (s3d_ffmlp_backward: colour head on color_in, density head on h0), so a
 * training forward takes the same launch.  Same arithmetic and rounding points as s3d_ffmlp_inference(density head) +
 * s3d_ffmlp_inference(colour head).  hidden_dim 64, ReLU, 16-column outputs.  enc_color (optional) fp16 level-major [16][B][2]:
 * the network Seal-3D trains (nerf/network.py:99-128) — the colour network then takes the 64-wide row of s3d_ngp_mid2_forward,
 * [half(SH_4(d)) | h1..h15 | enc_color | 0], built on chip (weights_color [64*64 | ...], color_in [B, 64]). */
int s3d_ffmlp_ngp_pair_inference(const uint16_t* inputs, const uint16_t* weights_sigma, const uint16_t* weights_color,
                                 uint32_t B, uint32_t hidden_dim, uint32_t num_layers_sigma, uint32_t num_layers_color,
                                 int input_layout, const int32_t* n_valid, const float* dirs, float* sigma, float* rgb,
                                 uint16_t* color_in, uint16_t* h0, const uint16_t* enc_color, s3d_stream_t stream);
/* ffmlp.h:11; grad_weights fp16 [same layout as weights]: every element is written (accumulate_grad_weights = 0,
 * the reference zero-fills it first, ffmlp.py:72) or added to (accumulate_grad_weights = 1).
 * workspace: fp32 accumulation of the weight gradient (s3d_ffmlp_backward_workspace_size).
 * forward_buffer == backward_buffer == NULL selects the fused backward: the activations are re-computed from
 * `inputs` inside one kernel that also forms the data and weight gradients, so a training forward may skip
 * forward_buffer altogether (call s3d_ffmlp_forward with forward_buffer = NULL).  Shapes it covers:
 * s3d_ffmlp_fused_backward_supported() != 0 (hidden 32/64, <= 3 hidden matrices, no sine).
 * found_inf (optional, build extension): device float raised to 1 when a written grad_weights element is non-finite
 * (GradScaler's check made by the kernel that writes the gradient).  Never cleared here. */
int s3d_ffmlp_fused_backward_supported(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                                       uint32_t num_layers, uint32_t activation);
size_t s3d_ffmlp_backward_workspace_size(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                                         uint32_t num_layers);
int s3d_ffmlp_backward(const uint16_t* grad, const uint16_t* inputs, const uint16_t* weights,
                       const uint16_t* forward_buffer, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                       uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                       uint32_t output_activation, int calc_grad_inputs, uint16_t* backward_buffer,
                       uint16_t* grad_inputs, uint16_t* grad_weights, void* workspace,
                       size_t workspace_bytes, int input_layout, int accumulate_grad_weights,
                       const int32_t* n_valid, float* found_inf, const float* grad_rgb, const float* rgb_head,
                       const float* mid_grad_sigma, const uint16_t* mid_grad_color_in, const uint16_t* mid_h0,
                       s3d_stream_t stream);
/* Build extension: s3d_ffmlp_backward with accumulate_grad_weights = 2 (fused backward only) leaves the per-workgroup partial
 * sums of the weight gradient in `workspace` and skips its reduce launch; this call finishes TWO such networks (the colour and
 * the density network of one step, each with its own workspace) in one launch: grad_weights_x fp16 written (accumulate_x = 0)
 * or added to (1), found_inf_x raised like s3d_ffmlp_backward's.  B_x etc.: the arguments of the backward call it finishes. */
int s3d_ffmlp_wgrad_reduce_pair(const void* workspace_a, uint32_t B_a, uint32_t input_dim_a, uint32_t hidden_dim_a,
                                uint32_t num_layers_a, uint16_t* grad_weights_a, int accumulate_a, float* found_inf_a,
                                const void* workspace_b, uint32_t B_b, uint32_t input_dim_b, uint32_t hidden_dim_b,
                                uint32_t num_layers_b, uint16_t* grad_weights_b, int accumulate_b, float* found_inf_b,
                                s3d_stream_t stream);
/* ffmlp.h:13-14: the reference allocates split-K side streams here; this build fuses the weight
 * gradient into the backward launch sequence on the caller's stream, so these are no-ops kept for
 * surface compatibility. */
int s3d_ffmlp_allocate_splitk(size_t n);
int s3d_ffmlp_free_splitk(void);

/* ------------------------------------------------------------------ TensoRF vector-matrix features
 * tensoRF/network.py:112-153 get_sigma_feat / get_color_feat (12 F.grid_sample calls, bilinear, zeros padding,
 * align_corners=True, + stack / cat / mul / sum) in one pass, csrc/tensorf.hip.
 * x [N,3] fp32 in [-1,1] (normalised like network.py:160); planes / lines / rank / resolution: HOST arrays of 3:
 * planes[i] device fp32 [rank[i], res[m1], res[m0]] with (m0,m1) = mat_ids[i] = (0,1),(0,2),(1,2); lines[i] device fp32
 * [rank[i], res[vec_ids[i]]], vec_ids = 2,1,0 (network.py:37-38, :105-106).
 * reduce = 1: out [N] = sum_i sum_r plane*line (sigma_feat); reduce = 0: out [rank0+rank1+rank2, N] = the products
 * (the tensor the reference transposes into basis_mat, network.py:147). */
int s3d_vm_features_forward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                            const uint32_t* rank, const uint32_t* resolution, int reduce, float* out,
                            const float* const* planes_t, const float* const* lines_t, const int32_t* n_valid,
                            s3d_stream_t stream);
/* Parameter gradients of the same op (what autograd derives from the grid_sample calls: F.grid_sample's backward
 * scatter-adds every corner with a global atomic).  Binned instead: s3d_vm_backward_keys writes keys [6,N] i32 (rows 0-2
 * the 8x8-cell plane tile of component i, rows 3-5 its 64-row line chunk; 0x7fffffff = contributes nothing); the caller
 * sorts each row, passes perm [6,N] i32 (point ids in key order) and start [6,n_bounds] i32 (first sorted position with
 * key >= t, n_bounds > s3d_vm_backward_max_bins(resolution) + 1), a scratch gm [N, sum rank] (no initialisation needed) and
 * zero-initialised gradient buffers shaped like the factors.  grad: [N] (reduce = 1) or [N, sum rank] (reduce = 0: the
 * gradient of the products, point-major — the memory layout autograd hands back through the reference's `.T`).
 * bound_words: four zero-initialised device words per call — the kernels sum a tile's contributions in LDS as 64-bit fixed
 * point (integer LDS atomics retire ~10x faster than float ones on MI355X) and keep the bounds that fix its scale there
 * (max |grad|, max |line parameter|, max |g m|, max column sum of |basis|); a non-finite bound poisons the gradients.
 * line_scratch: sum_i rank_i * resolution[vec_ids[i]] floats, no initialisation: the bound pass leaves the line factors there
 * transposed ([Dn][rank]) for the plane pass, whose lanes are rank channels.
 * found_inf (optional device float): set to 1 when a bound is not finite — the condition under which these kernels write a
 * non-finite gradient — so a GradScaler need not read the 69 MB of factor gradients again to find out.
 * planes_t / lines_t (optional, round 6): rank-fastest shadows of the factors — planes_t[i] [H][W][rank_i], lines_t[i] [Dn][rank_i],
 * 16-byte aligned, written by s3d_vm_transpose_factors from the current parameters (the caller refreshes them when the parameters
 * change).  With them a corner's rank channels are one contiguous run instead of rank_i words H * W * 4 bytes apart: the forwards
 * (ranks that are multiples of four) read four ranks per 16-byte load, the plane passes of the backward load a tile's window as 81
 * runs.  Same values, same arithmetic, same results bit for bit; NULL: the parameters' own layout.
 * n_valid (optional, round 6; every s3d_vm_* entry point that walks the points, and s3d_freq_encode_pack_*): the device-side
 * sample count of a padded batch — rows [round_up(*n_valid, 128), N) are absent: the forwards do not write them, the binning
 * sorts them behind every bin (the backward passes then never see them) and the bound pass does not look at their gradients.
 * stage / stage_bytes (optional, round 6): s3d_vm_backward_stage_bytes(N, rank, resolution) bytes of 16-byte aligned device
 * scratch, no initialisation.  With it the cells several workgroups add to (the border of a whole tile's 9x9 window, a line
 * chunk's 65 cells) leave as plain stores into per-tile / per-workgroup rows and one extra launch adds the rows in a FIXED
 * order; NULL (or too small): those cells take global atomics, as before.  Same sums within fp32 summation order. */
size_t s3d_vm_backward_stage_bytes(uint32_t N, const uint32_t* rank, const uint32_t* resolution);
int s3d_vm_transpose_factors(const float* const* planes, const float* const* lines, const uint32_t* rank, const uint32_t* resolution,
                             float* const* planes_t, float* const* lines_t, s3d_stream_t stream);
uint32_t s3d_vm_backward_max_bins(const uint32_t* resolution);
int s3d_vm_backward_keys(const float* x, uint32_t N, const uint32_t* rank, const uint32_t* resolution, int32_t* keys,
                         s3d_stream_t stream);
/* The whole binning in one call: perm [6,N] i32 and start [6,n_bounds] i32 (n_bounds >= max_bins + 2) as described above, by a
 * counting sort (keys + wave-aggregated bin counts, scan, scatter); the order of the points inside a bin is unspecified (the
 * backward kernels sum a bin exactly, in fixed point).  workspace: s3d_vm_backward_bins_workspace_size(N, n_bounds) bytes. */
size_t s3d_vm_backward_bins_workspace_size(uint32_t N, uint32_t n_bounds);
int s3d_vm_backward_bins(const float* x, uint32_t N, const uint32_t* rank, const uint32_t* resolution, int32_t* perm,
                         int32_t* start, uint32_t n_bounds, void* workspace, size_t workspace_bytes, const int32_t* n_valid,
                         s3d_stream_t stream);
int s3d_vm_features_backward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                             const uint32_t* rank, const uint32_t* resolution, int reduce, const float* grad,
                             const int32_t* perm, const int32_t* start, uint32_t n_bounds, float* gm,
                             float* const* grad_planes, float* const* grad_lines, uint32_t* bound_words, float* line_scratch,
                             void* stage, size_t stage_bytes, float* found_inf, const float* const* planes_t, const int32_t* n_valid,
                             s3d_stream_t stream);

/* The colour features with basis_mat applied inside the kernel (tensoRF/network.py:149-153: `basis_mat((mat * vec).T)`, an
 * nn.Linear(sum rank, basis_rows, bias=False) that runs under fp16 autocast): out [N, basis_rows] fp16 =
 * half(sum_row half(basis[c][row]) * half(product[row][n])) with fp32 accumulation; the [sum rank, N] products are never
 * written.  basis: fp16 [basis_rows, sum rank] (the Linear's weight), basis_rows <= 32.
 * s3d_vm_color_backward: from grad_out fp16 [N, 32] (the gradient of that output, rows zero-padded to 32 columns = 64 bytes,
 * 16-byte aligned) the factor gradients as
 * s3d_vm_features_backward writes them (same perm / start / gm / zero-initialised buffers) and grad_basis fp32
 * [basis_rows, sum rank] (zero-initialised; accumulated with atomics). */
int s3d_vm_color_forward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                         const uint32_t* rank, const uint32_t* resolution, const uint16_t* basis, uint32_t basis_rows,
                         uint16_t* out, const float* const* planes_t, const float* const* lines_t, const int32_t* n_valid,
                         s3d_stream_t stream);
int s3d_vm_color_backward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                          const uint32_t* rank, const uint32_t* resolution, const uint16_t* basis, uint32_t basis_rows,
                          const uint16_t* grad_out, const int32_t* perm, const int32_t* start, uint32_t n_bounds,
                          float* gm, float* const* grad_planes, float* const* grad_lines, float* grad_basis,
                          uint32_t* bound_words, float* line_scratch, void* stage, size_t stage_bytes, float* found_inf,
                          const float* const* planes_t, const int32_t* n_valid, s3d_stream_t stream);

/* Build extensions for the TensoRF step (chains of tiny launches otherwise):
 * s3d_aabb_normalize: out[n][a] = 2 (x[n][a] - aabb[a]) / (aabb[3 + a] - aabb[a]) - 1 (tensoRF/network.py:155-157, the reference's
 *   operation order; aabb = 6 DEVICE floats);
 * s3d_weighted_abs_sum: *out = sum_i weights[i] * sum |tensors[i]| over up to 8 fp32 tensors (host arrays of device pointers /
 *   element counts / weights) — density_loss() of tensoRF/network.py:259-263 with weights 1 / numel; workspace:
 *   s3d_weighted_abs_sum_workspace_size() bytes of scratch; fixed summation order. */
int s3d_aabb_normalize(const float* x, const float* aabb, uint32_t N, float* out, s3d_stream_t stream);
size_t s3d_weighted_abs_sum_workspace_size(void);
int s3d_weighted_abs_sum(const float* const* tensors, const uint64_t* numel, const float* weights, int32_t count, float* out,
                         float* workspace, s3d_stream_t stream);

/* s3d_pack_linear_chain: the fp32 [rows_i, cols_i] weights of a bias-free nn.Linear chain into the flat fp16 vector s3d_ffmlp_*
 * take ([W, in_pad] | (n - 1) x [W, W] | [16, W]: matrix i at its running offset, row stride ld[i] >= cols[i], padded_rows[i] >=
 * rows[i] rows, padding zero) in one launch; s3d_unpack_linear_chain: the flat fp16 gradient back into fp32 matrices of the
 * parameters' shapes.  Host arrays of `count` <= 8 entries. */
int s3d_pack_linear_chain(const float* const* mats, const uint32_t* rows, const uint32_t* cols, const uint32_t* padded_rows,
                          const uint32_t* ld, int32_t count, uint16_t* flat, s3d_stream_t stream);
int s3d_unpack_linear_chain(const uint16_t* flat, float* const* mats, const uint32_t* rows, const uint32_t* cols,
                            const uint32_t* padded_rows, const uint32_t* ld, int32_t count, s3d_stream_t stream);

/* ------------------------------------------------------------------ NGP head glue
 * The elementwise steps between the two MLPs of nerf/network_ff.py:55-96 (slice / trunc_exp / SH / cat / cast /
 * sigmoid and their backward nodes) as two streaming kernels per direction, csrc/ngp_head.hip.
 * h, color_in, out and their gradients are fp16 row-major ([B,16], [B,32], [B,16]); sigma, rgb, dirs fp32. */
int s3d_ngp_mid_forward(const uint16_t* h, const float* dirs, uint32_t B, float* sigma, uint16_t* color_in,
                        const int32_t* n_valid, s3d_stream_t stream);
int s3d_ngp_mid_backward(const uint16_t* grad_color_in, const float* grad_sigma /* or NULL */, const uint16_t* h,
                         uint32_t B, uint16_t* grad_h, const int32_t* n_valid, s3d_stream_t stream);
/* Two-encoder network (nerf/network.py:99-128 — the net Seal-3D trains): colour-net input [B,64] =
 * [half(SH_4(d)) | h1..h15 | encoder_color(x) (32) | 0]; enc_color and grad_enc_color (may be NULL) are level-major
 * [16][B][2] fp16, the grid kernels' own layout. */
int s3d_ngp_mid2_forward(const uint16_t* h, const float* dirs, const uint16_t* enc_color, uint32_t B, float* sigma,
                         uint16_t* color_in, const int32_t* n_valid, s3d_stream_t stream);
int s3d_ngp_mid2_backward(const uint16_t* grad_color_in, const float* grad_sigma /* or NULL */, const uint16_t* h,
                          uint32_t h_stride /* 16: h is the density network's [B, 16] output; 1: its first column [B] alone */,
                          uint32_t B, uint16_t* grad_h, uint16_t* grad_enc_color /* or NULL */, const int32_t* n_valid,
                          s3d_stream_t stream);
int s3d_ngp_rgb_forward(const uint16_t* out, uint32_t B, float* rgb, const int32_t* n_valid, s3d_stream_t stream);
int s3d_ngp_rgb_backward(const float* grad_rgb, const float* rgb, uint32_t B, uint16_t* grad_out,
                         const int32_t* n_valid, s3d_stream_t stream);
/* Loss head of one ray batch: loss = mean((image + (1 - weights_sum) * bg - gt)^2)  (nerf/renderer.py:316 background
 * compositing + nerf/utils.py:484 MSE); bg_rgb = 3 HOST floats; loss / grad_loss are single device floats.
 * s3d_bg_mse_forward with grad_loss != NULL also writes what s3d_bg_mse_backward would for that upstream gradient (under
 * loss scaling the loss's upstream gradient is the scale, known before the backward pass): one launch instead of two.
 * depth / gt_depth [N] (optional, both or neither) add Seal-3D's depth term depth_weight * mean(|nan_to_num(depth) - gt_depth|)
 * (nerf/utils.py:486-489) to the VALUE of the loss; like the reference's composite backward (raymarching.py:274: "grad_depth is
 * not used now") it contributes no gradient. */
int s3d_bg_mse_forward(const float* image, const float* weights_sum, const float* gt, const float* bg_rgb, uint32_t N,
                       float* loss, const float* grad_loss, float* grad_image, float* grad_weights_sum,
                       const float* depth, const float* gt_depth, float depth_weight, s3d_stream_t stream);
int s3d_bg_mse_backward(const float* image, const float* weights_sum, const float* gt, const float* bg_rgb, uint32_t N,
                        const float* grad_loss, float* grad_image, float* grad_weights_sum, s3d_stream_t stream);
/* Build extension — targets of a teacher-rendered ray batch (SealNeRF/trainer.py:506-586 `proxy_truth`): out_rgb [N,3] =
 * nan_to_num(image + (1 - weights_sum) * bg) (nerf/renderer.py:316), out_depth [N] (optional) = nan_to_num(depth), nan -> 0 and
 * +-inf -> +-FLT_MAX as torch.nan_to_num(nan=0.); bg_rgb = 3 HOST floats. */
int s3d_bg_targets(const float* image, const float* weights_sum, const float* depth, const float* bg_rgb, uint32_t N,
                   float* out_rgb, float* out_depth, s3d_stream_t stream);
/* Build extension — Seal-3D's local-pretraining loss on one point chunk (SealNeRF/trainer.py:455-469: L1Loss(sigma) +
 * L1Loss(colour), means): loss = sum|sigma - gt_sigma| / n_total + sum|color - gt_color| / (3 n_total); n_total >= n is the
 * size of the whole chunk when the call sees one rank's shard of it; sigma / color (and the gradients) hold n_rows >= n rows,
 * rows [n, n_rows) being padding of the prediction (no term, zero gradient; the targets hold n rows).  grad_loss != NULL (a device float, the loss scale):
 * grad_sigma [n] and grad_color [n,3] = sign(.) * *grad_loss / n_total resp. / (3 n_total) are written by the same launch.
 * workspace: s3d_l1_pair_workspace_size() bytes, zero-filled ONCE by the caller (the kernel leaves its ticket word zero);
 * the value is summed in a fixed order (reproducible). */
size_t s3d_l1_pair_workspace_size(void);
int s3d_l1_pair_loss(const float* sigma, const float* color, const float* gt_sigma, const float* gt_color, uint32_t n,
                     uint32_t n_rows, uint32_t n_total, float* loss, const float* grad_loss, float* grad_sigma, float* grad_color,
                     void* workspace, s3d_stream_t stream);

/* ------------------------------------------------------------------ Seal proxy mapper (bbox tool)
 * SealNeRF/seal_utils.py:132-153 (map_mask), :237-279 (map_to_origin), :630-685 (two-ray Moller-Trumbore inside test)
 * in one pass over the sample points.  points/dirs/out_* are DEVICE [M,3] f32, mask DEVICE [M] u8; everything that
 * describes the edit is HOST data (a handful of constants): triangles [n_tris,3,3], bounds [n_bounds,2,3] (min,max),
 * inv_transform [4,4] row-major, inv_rotation [3,3], inv_scale [3], center [3]; empty_bound [2,3] + map_source [3] or
 * both NULL.  out_dirs/dirs may both be NULL. */
int s3d_seal_bbox_map(const float* points, const float* dirs, uint32_t M, const float* triangles, uint32_t n_tris,
                      const float* bounds, uint32_t n_bounds, const float* inv_transform, const float* inv_rotation,
                      const float* inv_scale, const float* center, const float* empty_bound, const float* map_source,
                      float* out_points, float* out_dirs, uint8_t* mask, const int32_t* n_valid /* padded sample batch: rows past it are skipped, or NULL */,
                      s3d_stream_t stream);

/* Colour edit of the bbox tool applied to the samples the proxy moved — `rgbs[mask] = map_color(.., rgbs[mask])` of
 * SealNeRF/renderer.py:316, 396-399 with seal_utils.py:48-58 (map_color), :739-769 (modify_hsv, modify_rgb) and
 * color_utils.py:33-66 (rgb2hsv_torch / hsv2rgb_torch).  rgbs / out DEVICE [M,3] f32 or f16 (`dtype`; out may alias rgbs),
 * mask DEVICE [M] u8 (s3d_seal_bbox_map's).  hsv: HOST [3] offsets added to (h, s, v), or NULL.  rgb_target: HOST [3] target
 * colour or NULL — every moved sample takes the target's hue and saturation and the value
 * clamp(target_v + (v_i - mean_moved(v)) + light_offset, 0, 1); the mean is taken over the moved rows of THIS call (a batch
 * statistic, as in the reference) in `stats`, DEVICE 16 bytes (order-independent fixed-point sum + count; cleared by the call).
 * With both, hsv is applied first and its RGB result converted again (the reference's two steps).  Rows where mask == 0 are
 * copied. */
int s3d_seal_map_color(const void* rgbs, const uint8_t* mask, uint32_t M, int dtype, const float* hsv, const float* rgb_target,
                       float light_offset, void* out, void* stats, const int32_t* n_valid /* or NULL */, s3d_stream_t stream);

/* ------------------------------------------------------------------ parameter update
 * The reference's update is torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15) under torch.cuda.amp.GradScaler
 * (nerf/utils.py:356-361, 495-537; main_SealNeRF.py:283-288).  These three calls are that update taken directly
 * from the gradient the backward kernels produced (fp16 for the hash tables), see csrc/optim.hip.
 * found_inf / grad_scale / step are single device floats (GradScaler's found_inf and scale, Adam's step count).
 * s3d_grads_nonfinite sets *found_inf = 1 if any element is inf/NaN (never clears it).
 * s3d_adam_step updates param / exp_avg / exp_avg_sq (fp32) with grad / *grad_scale as step number *step + 1, and
 * does nothing when *found_inf != 0; param_half (optional) receives the fp16 copy of the updated parameters.
 * s3d_adam_advance increments *step unless *found_inf != 0 (call once per optimizer step, after the tensors).
 * lr_scale (optional, device): the update uses lr * *lr_scale — the factor of a learning-rate schedule
 * (torch.optim.lr_scheduler.LambdaLR in main_SealNeRF.py:283-288), read at run time so that a step captured in a HIP graph
 * follows the schedule without being re-captured. */
int s3d_grads_nonfinite(const void* grad, size_t n, int dtype, float* found_inf, s3d_stream_t stream);
int s3d_adam_step(float* param, const void* grad, int grad_dtype, float* exp_avg, float* exp_avg_sq,
                  uint16_t* param_half, size_t n, float lr, float beta1, float beta2, float eps,
                  const float* step, const float* grad_scale, const float* found_inf, const float* lr_scale,
                  s3d_stream_t stream);
int s3d_adam_advance(float* step, const float* found_inf, s3d_stream_t stream);
/* s3d_adam_step for every tensor of the optimizer in one launch (same arithmetic per element).  consume_grads != 0: every
 * gradient is cleared behind the read — also on a skipped (*found_inf != 0) step — for producers that accumulate into it. */
typedef struct s3d_adam_tensor {
    float* param;
    void* grad; /* written only with consume_grads */
    float* exp_avg;
    float* exp_avg_sq;
    uint16_t* param_half; /* optional fp16 copy of the updated parameters */
    size_t n;
    float lr, beta1, beta2, eps;
    int grad_dtype; /* S3D_F32 or S3D_F16 */
    int consume;    /* != 0: this tensor's gradient is cleared behind the read (as consume_grads, per tensor) */
    /* pack_stride != 0: `grad` and `param_half` are views into a PACKED weight buffer — element (r, c) of the [n / pack_cols,
     * pack_cols] parameter lives at r * pack_stride + c (the nn.Linear weights of nerf/network.py inside the fused MLP
     * kernels' padded [out, in] layout); param / exp_avg / exp_avg_sq stay contiguous.  0: contiguous (the reference case). */
    uint32_t pack_cols, pack_stride;
    /* Build extension: != 0 adds the gradient of the penalty l1 * sum|param| to the (unscaled) gradient inside the update,
     * l1 * sign(param) with sign(0) = 0 — TensoRF's L1 term on the density factors (tensoRF/network.py:259-263, tensoRF/utils.py:
     * 42-49: l1 = l1_reg_weight / numel) without the sign / scale / accumulate passes over the factors. */
    float l1;
} s3d_adam_tensor;
int s3d_adam_step_multi(const s3d_adam_tensor* tensors /* host array */, int32_t n_tensors, const float* step,
                        const float* grad_scale, const float* found_inf, const float* lr_scale, int consume_grads,
                        s3d_stream_t stream);
/* GradScaler.update(): scale *= backoff on overflow, *= growth after growth_interval clean steps; clears *found_inf.
 * adam_step (optional): s3d_adam_advance of that step count folded into the same launch (before the flag is cleared). */
int s3d_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, float growth_factor,
                      float backoff_factor, int32_t growth_interval, float* adam_step, s3d_stream_t stream);
/* Build extension (graph-replayed step): the renderer keeps the marcher's {samples, rays} counters of the last 16 training
 * steps (nerf/renderer.py:106, 352-356: `step_counter[local_step % 16]`).  Device-side equivalent of that bookkeeping:
 * slot = cursor[0]; loss_ring[slot] = *loss (both optional); counter_ring[slot] = counter[0..1]; counter[0..1] = 0;
 * cursor[0] = (slot + 1) % ring; cursor[1] += 1 (running step number, the `noise_step` of s3d_near_far_from_aabb).
 * loss_slots > 0: loss_ring has loss_slots entries of its own and the loss is filed at cursor[1] % loss_slots (before the
 * increment) — a longer history than the counter ring, so that a loss handed out as a view stays valid for loss_slots steps. */
int s3d_step_ring_push(const float* loss, int32_t* counter, float* loss_ring, int32_t* counter_ring, int32_t* cursor,
                       int32_t ring, int32_t loss_slots, s3d_stream_t stream);
/* s3d_scaler_update followed by s3d_step_ring_push, one launch. */
int s3d_step_epilogue(float* scale, int32_t* growth_tracker, float* found_inf, float growth_factor, float backoff_factor,
                      int32_t growth_interval, float* adam_step, const float* loss, int32_t* counter, float* loss_ring,
                      int32_t* counter_ring, int32_t* cursor, int32_t ring, int32_t loss_slots, s3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEAL3D_HIP_H */
