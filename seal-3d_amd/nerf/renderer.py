"""NeRFRenderer — the build's counterpart of nerf/renderer.py (the caller of the hot path).

Reproduces the behaviour of `run` (:125-253, the sampling path without the occupancy grid) and `sample_pdf` (:12-46),
`run_cuda` (training branch :256-321, inference loop :323-372),
`update_extra_state` (:444-538), `mark_untrained_grid` (:379-441) and `reset_extra_state` on top of the
drop-in `raymarching` package.  Buffers and attribute names are the reference's (`density_grid`,
`density_bitfield`, `step_counter`, `mean_count`, `mean_density`, `iter_density`, `local_step`) so
checkpoints keep their keys.

MI355X additions (same results): `device_compaction=True` replaces the per-iteration host boolean-mask
`rays_alive[rays_alive >= 0]` by a wave-ballot compaction kernel (one 4-byte D2H read per iteration instead
of an implicit sync + index kernel chain).
"""
import math

import numpy as np
import torch
import torch.nn as nn

import raymarching
import s3d_hip


import contextlib

_null_context = contextlib.nullcontext


def _meshgrid(*args):
    return torch.meshgrid(*args, indexing="ij")


def sample_pdf(bins, weights, n_samples, det=False):
    """inverse-CDF sampling of `n_samples` depths per ray from the piecewise-constant density `weights` over `bins`
    (nerf/renderer.py:12-46): bins [B, T], weights [B, T - 1] -> [B, n_samples]"""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples).to(weights.device)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples]).to(weights.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    pick = torch.stack([below, above], -1)
    shape = [pick.shape[0], pick.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, pick)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, pick)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1,
                 device_compaction=True, infer_batch_scale=1):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        self.device_compaction = device_compaction
        # inference loop: the host reads the alive-ray count back only every `sync_every` iterations; in between, the kernels
        # take the count from device memory and the launch geometry / buffers are sized by the last known count (an upper
        # bound).  1 = a read-back per iteration, as the reference's boolean-mask compaction implies.
        self.sync_every = 4
        # inference: samples marched per alive ray and iteration = min(scale*N // n_alive, 8*scale).  1 = the reference's
        # heuristic (keeps ~N samples per iteration, nerf/renderer.py:352); larger values trade a few wasted samples of
        # rays that terminate mid-chunk for fewer, fuller iterations.  The composited result does not depend on it.
        self.infer_batch_scale = infer_batch_scale

        aabb = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())

        # data parallelism (parallel/dist.py): when set, the density queries of the occupancy sweep are split over the ranks
        # and all-gathered, and the sweep draws its random numbers from a generator every rank seeds alike, so
        # all replicas end up with bit-identical grids without a broadcast
        self.dist_shard = None
        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    def honours_row_limit(self, rows):
        """True when forward() on a [rows, 3] batch takes the announced device-side sample count (s3d_hip.row_limit) on its
        whole path — then run_cuda keeps un-budgeted sample buffers at their full static extent instead of reading the
        count back to trim them"""
        return False

    def honours_row_limit_under_autocast(self, rows):
        with torch.autocast("cuda", dtype=torch.float16):
            return bool(self.honours_row_limit(rows))

    def _plain_sample_path(self):
        """no per-sample hook between the marcher and the network (a hook may read sample rows as whole tensors)"""
        return type(self).map_samples is NeRFRenderer.map_samples and type(self).map_colors is NeRFRenderer.map_colors

    # ---- per-sample hooks (identity here; the Seal teacher overrides them, SealNeRF/renderer.py:291-316, 381-399)
    def map_samples(self, xyzs, dirs):
        return xyzs, dirs, None

    def map_colors(self, xyzs, dirs, rgbs, mask):
        return rgbs

    # ---- to be provided by the network subclass
    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    def _batch_dependent_colors(self):
        return False

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------ run_cuda
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 T_thresh=1e-4, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device

        aabb = self.aabb_train if self.training else self.aabb_infer
        noises = None
        noise_step = getattr(self, "_noise_step", None) if (self.training and perturb) else None
        if noise_step is not None:  # graph-replayed step (nerf/trainer.py): near / far and the jitter are made by the marcher
            nears = fars = None
        else:
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)

        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(sph, rays_d)
        elif bg_color is None:
            bg_color = 1

        results = {}
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            if not getattr(self, "_counter_prezeroed", False):  # (a graph-replayed step clears it behind itself, nerf/trainer.py)
                counter.zero_()
            self.local_step += 1
            budgeted = (not force_all_rays) and self.mean_count > 0
            trim = budgeted or not self.honours_row_limit(N * max_steps)  # (no budget: N * max_steps rows, 128-aligned)
            # every consumer of the sample rows takes the device-side count: nothing reads behind it, the HIP kernels keep the
            # unfilled rows in front of it zero themselves -> no zero fill of the M-row buffers (forward and gradients)
            M_rows = self.mean_count + 128 - self.mean_count % 128 if budgeted else N * max_steps  # (the marcher's M)
            lean = rays_o.is_cuda and (budgeted or not trim) and self.honours_row_limit(M_rows) and self._plain_sample_path()
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(
                rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, counter,
                self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps,
                *(() if trim and noises is None and not lean and noise_step is None else
                  (trim, noises, not lean) + (() if noise_step is None else
                                              (aabb, self.min_near, noise_step, getattr(self, "_noise_key", 0)))))
            # the buffers are padded to M rows (raymarching.py:205-207); counter[0] says on the device how many hold samples
            # (a proxy mapper skips the rows behind the count only when the network behind it skips them too: otherwise the
            #  network would read rows the mapper never wrote)
            if self.honours_row_limit(xyzs.shape[0]):
                with s3d_hip.row_limit(counter, xyzs.shape[0]):
                    mxyzs, mdirs, mmask = self.map_samples(xyzs, dirs)
                    sigmas, rgbs = self(mxyzs, mdirs)
                    rgbs = self.map_colors(mxyzs, mdirs, rgbs, mmask)  # (the mask's rows behind the count were never written)
            else:
                mxyzs, mdirs, mmask = self.map_samples(xyzs, dirs)
                with s3d_hip.row_limit(counter, xyzs.shape[0]):
                    sigmas, rgbs = self(mxyzs, mdirs)
                rgbs = self.map_colors(mxyzs, mdirs, rgbs, mmask)
            if self.density_scale != 1:  # (x1 is the identity: skip the pass over [M])
                sigmas = self.density_scale * sigmas
            fused = kwargs.get("fused_loss")  # (nerf/trainer.py: the criterion and its gradient inside the compositing launch)
            if (fused is not None and kwargs.get("defer_background", False) and not torch.is_tensor(bg_color)
                    and sigmas.is_cuda and fused.get("expected_grad") is not None
                    # (the one-launch composite + criterion exists on the product's composite path only: an A/B run on
                    #  another path takes the unfused sequence below instead of failing)
                    and hasattr(raymarching.raymarching._backend, "composite_rays_train_loss")
                    and getattr(raymarching.raymarching._backend, "_composite_path", 0) == 0):
                bg3 = (float(bg_color),) * 3 if not isinstance(bg_color, (tuple, list)) else tuple(float(v) for v in bg_color)
                results["loss"], weights_sum, depth, image = raymarching.composite_rays_train_loss(
                    sigmas, rgbs, deltas, rays, T_thresh, fused["gt"], bg3, fused["expected_grad"], fused.get("workspace"),
                    fused.get("gt_depth"), fused.get("depth_weight", 1.0), not lean)
            else:
                weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh,
                                                                             *(() if not lean else (False,)))
            if kwargs.get("defer_background", False) and not torch.is_tensor(bg_color):
                # the caller composites the background inside its fused loss kernel (nerf/trainer.py:bg_mse_loss)
                results["premultiplied"] = True
                results["bg_color"] = bg_color
            else:
                image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            results["weights_sum"] = weights_sum
        else:
            dtype = torch.float32  # outputs stay fp32; only the network runs in half under autocast
            weights_sum = torch.zeros(N, dtype=dtype, device=device)
            depth = torch.zeros(N, dtype=dtype, device=device)
            image = torch.zeros(N, 3, dtype=dtype, device=device)
            n_alive = N
            rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            use_dev_compaction = self.device_compaction and device.type == "cuda"
            # a proxy colour edit whose result depends on WHICH samples share a network batch (Seal's `rgb` edit keeps each
            # sample's brightness offset from the batch mean, seal_utils.py:753-769) pins the loop to the reference's shape:
            # exact alive count every iteration, n_step = max(min(N // n_alive, 8), 1)
            exact_batches = self._batch_dependent_colors()
            batch_scale = 1 if exact_batches else self.infer_batch_scale
            sync_free = use_dev_compaction and self.sync_every > 1 and not exact_batches
            # sync-free variant: `n_alive` is the host's upper bound, `cnt` (device) the real count; every kernel of the
            # iteration takes the count from the device, the bound is refreshed every `sync_every` iterations
            cnt = torch.full((1,), N, dtype=torch.int32, device=device) if sync_free else None
            rows = torch.zeros(1, dtype=torch.int32, device=device) if sync_free else None
            step = it = 0
            while step < max_steps and n_alive > 0:
                n_step = max(min(batch_scale * N // n_alive, 8 * batch_scale), 1)
                if sync_free:
                    xyzs, dirs, deltas = raymarching.march_rays(
                        n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                        self.grid_size, nears, fars, 128, perturb if step == 0 else False, dt_gamma, max_steps, cnt, rows)
                else:
                    xyzs, dirs, deltas = raymarching.march_rays(
                        n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                        self.grid_size, nears, fars, 128, perturb if step == 0 else False, dt_gamma, max_steps)
                mxyzs, mdirs, mmask = self.map_samples(xyzs, dirs)
                # slots a ray did not fill stay zero (deltas == 0) and composite_rays never reads their sigma / rgb; with a
                # device-side count the rows behind the last alive ray are skipped altogether (row_limit -> n_valid)
                with s3d_hip.live_rows(deltas), s3d_hip.row_limit(rows, xyzs.shape[0]) if sync_free else _null_context():
                    sigmas, rgbs = self(mxyzs, mdirs)
                if self.density_scale != 1:
                    sigmas = self.density_scale * sigmas
                rgbs = self.map_colors(mxyzs, mdirs, rgbs, mmask)
                if sync_free:
                    raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth,
                                               image, T_thresh, cnt)
                    rays_alive, cnt = raymarching.compact_rays_alive(rays_alive, n_alive, cnt)
                    it += 1
                    if it % self.sync_every == 0:
                        n_alive = int(cnt.item())
                        rays_alive = rays_alive[:max(n_alive, 1)]
                else:
                    raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth,
                                               image, T_thresh)
                    if use_dev_compaction:
                        rays_alive, c = raymarching.compact_rays_alive(rays_alive, n_alive)
                        n_alive = int(c.item())
                        rays_alive = rays_alive[:n_alive]
                    else:
                        rays_alive = rays_alive[rays_alive >= 0]
                        n_alive = rays_alive.shape[0]
                step += n_step
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color

        results["depth"] = depth.view(*prefix)
        results["image"] = image.view(*prefix, 3)
        return results

    # ------------------------------------------------------------------ density grid maintenance
    def _cascade_geometry(self, cas):
        bound = min(2 ** cas, self.bound)
        return bound, bound / self.grid_size

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """cells no training camera sees get density -1 and are never sampled (nerf/renderer.py:379-441)"""
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        B = poses.shape[0]
        fx, fy, cx, cy = intrinsic
        dev = self.density_bitfield.device
        axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        count = torch.zeros_like(self.density_grid)
        poses = poses.to(dev)
        for xs in axis:
            for ys in axis:
                for zs in axis:
                    xx, yy, zz = _meshgrid(xs, ys, zs)
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    indices = raymarching.morton3D(coords).long()
                    world = (2 * coords.float() / (self.grid_size - 1) - 1).unsqueeze(0)
                    for cas in range(self.cascade):
                        bound, hgs = self._cascade_geometry(cas)
                        cas_world = world * (bound - hgs)
                        for head in range(0, B, S):
                            tail = min(head + S, B)
                            cam = cas_world - poses[head:tail, :3, 3].unsqueeze(1)
                            cam = cam @ poses[head:tail, :3, :3]
                            mask = ((cam[:, :, 2] > 0)
                                    & (torch.abs(cam[:, :, 0]) < cx / fx * cam[:, :, 2] + hgs * 2)
                                    & (torch.abs(cam[:, :, 1]) < cy / fy * cam[:, :, 2] + hgs * 2)).sum(0).reshape(-1)
                            count[cas, indices] += mask
        self.density_grid[count == 0] = -1

    def _sweep_generator(self):
        """None (torch's global generator, the reference's stream) or, with a sharded sweep, a generator seeded from the
        update count: the same coordinates and jitter on every rank"""
        if self.dist_shard is None:
            return None
        g = torch.Generator(device=self.density_grid.device)
        g.manual_seed(0x5EA13D + int(self.iter_density))
        return g

    def _rand(self, shape, gen, like):
        if gen is None:
            return torch.rand_like(like)
        return torch.rand(shape, generator=gen, dtype=like.dtype, device=like.device)

    def _randint(self, high, shape, gen, device, dtype=torch.int64):
        if gen is None:
            return torch.randint(0, high, shape, dtype=dtype, device=device)
        return torch.randint(0, high, shape, generator=gen, dtype=dtype, device=device)

    @torch.no_grad()
    def _query_cells(self, coords, cas, gen=None):
        """jittered cell centres of cascade `cas` -> density (nerf/renderer.py:468-482)"""
        bound, hgs = self._cascade_geometry(cas)
        xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
        cas_xyzs = xyzs * (bound - hgs)
        cas_xyzs += (self._rand(cas_xyzs.shape, gen, cas_xyzs) * 2 - 1) * hgs

        def query(x):
            return self.density(x)["sigma"].reshape(-1).detach() * self.density_scale
        if self.dist_shard is not None:
            return self.dist_shard.sharded_map(query, cas_xyzs)
        return query(cas_xyzs)

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """EMA-max density grid update + bitfield re-pack + mean sample count (nerf/renderer.py:444-538)"""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        gen = self._sweep_generator()
        tmp_grid = -torch.ones_like(self.density_grid)
        if self.iter_density < 16:  # full sweeps first
            axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            for xs in axis:
                for ys in axis:
                    for zs in axis:
                        xx, yy, zz = _meshgrid(xs, ys, zs)
                        coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                        indices = raymarching.morton3D(coords).long()
                        for cas in range(self.cascade):
                            tmp_grid[cas, indices] = self._query_cells(coords, cas, gen).to(tmp_grid.dtype)
        else:  # then H^3/4 uniform + H^3/4 occupied cells per cascade
            N = self.grid_size ** 3 // 4
            for cas in range(self.cascade):
                coords = self._randint(self.grid_size, (N, 3), gen, dev)
                indices = raymarching.morton3D(coords).long()
                occ = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                pick = self._randint(occ.shape[0], [N], gen, dev, torch.long)
                occ = occ[pick]
                occ_coords = raymarching.morton3D_invert(occ)
                indices = torch.cat([indices, occ], dim=0)
                coords = torch.cat([coords, occ_coords], dim=0)
                tmp_grid[cas, indices] = self._query_cells(coords, cas, gen).to(tmp_grid.dtype)

        valid = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid[valid] = torch.maximum(self.density_grid[valid] * decay, tmp_grid[valid])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1
        thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, thresh, self.density_bitfield)

        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    # -- the steady-state (partial) update again, without host syncs and with static shapes: one HIP graph for a trainer
    #    that replays its steps (nerf/trainer.py:GraphedTrainer).  Same update rule, same sampling law (H^3/4 uniform
    #    cells + H^3/4 uniform picks among the occupied cells per cascade, jittered, EMA-max, nerf/renderer.py:497-538 of
    #    the reference); the random stream differs from `update_extra_state` (which consumes the torch RNG exactly like
    #    the reference): occupied cells are drawn through a prefix sum + binary search instead of nonzero() + randint.
    @staticmethod
    def _sorted_uniform(N, dev):
        """N iid U[0, 1) draws in ASCENDING order, without a sort: normalised partial sums of N + 1 exponential spacings are
        distributed exactly like the order statistics of N uniforms.  The cells of an occupancy sweep are drawn through this
        (a set of iid picks does not care about its order), so their morton indices come out sorted: the density queries
        walk the volume along the Z-curve instead of jumping at random — the hash-grid gathers of neighbouring samples share
        cache lines (2 M-point sweep: ~2x faster encoder pass) and the scatter into the grid is monotone."""
        s = torch.cumsum(torch.empty(N + 1, dtype=torch.float64, device=dev).exponential_(), dim=0)
        return s[:N] / s[N]

    @torch.no_grad()
    def _pick_occupied(self, cas, N):
        """N uniform draws (with replacement) among the cells of cascade `cas` with density > 0, as ascending morton indices;
        no host sync, static shapes.  (No occupied cell at all: the reference's randint(0, 0) raises; this returns the last
        cell.)"""
        grid = self.density_grid[cas]
        csum = torch.cumsum(grid > 0, dim=0, dtype=torch.int32)
        pick = (self._sorted_uniform(N, grid.device) * csum[-1]).to(torch.int32)  # uniform in [0, #occupied)
        return torch.searchsorted(csum, pick, right=True).clamp_(max=grid.shape[0] - 1)  # the pick-th occupied cell

    @torch.no_grad()
    def partial_grid_update_device(self, decay=0.95):
        """density_grid <- EMA-max with fresh samples; returns mean(clamp(density_grid, 0)) as a device scalar"""
        dev = self.density_grid.device
        H3 = self.grid_size ** 3
        N = H3 // 4
        if dev.type == "cuda" and self.density_grid.dtype == torch.float32:
            # native sweep (csrc/raymarching.hip): cells + jittered positions in one launch, scatter / EMA-max / mean in four —
            # instead of ~40 elementwise torch kernels around the density query
            R = s3d_hip.RaymarchingBackend
            if getattr(self, "_sweep_step", None) is None or self._sweep_step.device != dev:
                self._sweep_step = torch.zeros(1, dtype=torch.int32, device=dev)
                self._sweep_key = torch.initial_seed() & 0xFFFFFFFF
            total = None
            for cas in range(self.cascade):
                bound, hgs = self._cascade_geometry(cas)
                grid = self.density_grid[cas]
                csum = torch.cumsum(grid > 0, dim=0, dtype=torch.int32)
                cells, xyzs = R.sweep_draw(self._sorted_uniform(N, dev), self._sorted_uniform(N, dev), csum, self.grid_size, bound,
                                           hgs, self._sweep_key + cas, self._sweep_step)
                sigma = self.density(xyzs)["sigma"].reshape(-1).detach()
                if sigma.dtype not in (torch.float16, torch.float32):
                    sigma = sigma.float()
                part = R.sweep_update(grid, cells, sigma.contiguous(), self.density_scale, decay,
                                      self._sweep_step if cas == self.cascade - 1 else None)
                total = part if total is None else total + part
            return total / self.density_grid.numel()
        tmp_grid = torch.full_like(self.density_grid, -1)
        for cas in range(self.cascade):
            # (uniform cells: a uniform morton index IS a uniform cell — the curve is a bijection of the H^3 grid)
            indices = (self._sorted_uniform(N, dev) * H3).long().clamp_(max=H3 - 1)
            coords = raymarching.morton3D_invert(indices)
            occ = self._pick_occupied(cas, N)
            occ_coords = raymarching.morton3D_invert(occ)
            indices = torch.cat([indices, occ], dim=0)
            coords = torch.cat([coords, occ_coords], dim=0)
            tmp_grid[cas, indices] = self._query_cells(coords, cas).to(tmp_grid.dtype)
        valid = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid.copy_(torch.where(valid, torch.maximum(self.density_grid * decay, tmp_grid), self.density_grid))
        # mean(clamp(grid, 0)) in two row-wise stages: torch.mean over 2M elements is a multi-block reduction whose
        # semaphores are cleared by a memset node, and memset nodes of a captured graph stop taking effect from the second
        # replay on (ROCm 7.2) — the mean read back 0 and the bitfield filled up
        g = self.density_grid.clamp(min=0)
        return g.view(-1, 4096).sum(dim=1).sum() / g.numel()

    @torch.no_grad()
    def finish_extra_state(self, mean_density_dev):
        """bitfield re-pack + mean sample count from the device results of `partial_grid_update_device`: ONE host read"""
        total_step = min(16, self.local_step)
        counted = self.step_counter[:max(total_step, 1), 0].sum().float()  # < 2^24: exact
        mean_density, count_sum = torch.stack([mean_density_dev.float().reshape(()), counted]).tolist()
        self.mean_density = mean_density
        self.iter_density += 1
        thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, thresh, self.density_bitfield)
        if total_step > 0:
            self.mean_count = int(count_sum / total_step)
        self.local_step = 0

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        """The sampling path without the occupancy grid (`cuda_ray` off; nerf/renderer.py:125-253): `num_steps` stratified
        samples between the ray's near and far, `upsample_steps` more drawn from the coarse weights (sample_pdf), alpha
        compositing with torch ops, colours evaluated only where the weight exceeds 1e-4.  BASELINE configs[0] renders its
        64x64 plumbing frame through this path (`num_steps=512`, main_SealNeRF.py:47).  The only native op is
        near_far_from_aabb (the reference's `run` calls the extension for it too, :141)."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        z_vals = torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand((N, num_steps))
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist

        def points(z):
            x = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
            return torch.min(torch.max(x, aabb[:3]), aabb[3:])

        def alpha_weights(z, sigma):
            d = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], dim=-1)
            alphas = 1 - torch.exp(-d * self.density_scale * sigma)
            shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
            return alphas * torch.cumprod(shifted, dim=-1)[..., :-1], d

        xyzs = points(z_vals)
        dens = {k: v.view(N, num_steps, -1) for k, v in self.density(xyzs.reshape(-1, 3)).items()}
        if upsample_steps > 0:
            with torch.no_grad():
                weights, deltas = alpha_weights(z_vals, dens["sigma"].squeeze(-1))
                z_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z = sample_pdf(z_mid, weights[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = points(new_z)
            new_dens = {k: v.view(N, upsample_steps, -1) for k, v in self.density(new_xyzs.reshape(-1, 3)).items()}
            z_vals, order = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], dim=1)
            xyzs = torch.gather(xyzs, dim=1, index=order.unsqueeze(-1).expand_as(xyzs))
            for k in dens:
                both = torch.cat([dens[k], new_dens[k]], dim=1)
                dens[k] = torch.gather(both, dim=1, index=order.unsqueeze(-1).expand_as(both))
        weights, _ = alpha_weights(z_vals, dens["sigma"].squeeze(-1))
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        dens = {k: v.view(-1, v.shape[-1]) for k, v in dens.items()}
        mask = weights > 1e-4
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **dens).view(N, -1, 3)
        weights_sum = weights.sum(dim=-1)
        depth = torch.sum(weights * ((z_vals - nears) / (fars - nears)).clamp(0, 1), dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(sph, rays_d.reshape(-1, 3))
        elif bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "weights_sum": weights_sum}

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """nerf/renderer.py:541-577: run_cuda with `cuda_ray`, else `run` — staged over ray chunks on request (never staged
        with cuda_ray, like the reference)"""
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, **kwargs)
        if not staged:
            return self.run(rays_o, rays_d, **kwargs)
        B, N = rays_o.shape[:2]
        depth = torch.empty((B, N), device=rays_o.device)
        image = torch.empty((B, N, 3), device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                part = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                depth[b:b + 1, head:tail] = part["depth"]
                image[b:b + 1, head:tail] = part["image"]
        return {"depth": depth, "image": image}
