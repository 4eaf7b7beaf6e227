"""Teacher -> student distillation steps of Seal-3D (SealNeRF/trainer.py), data-parallel over points / rays.

* local pretraining (`init_pretraining` :88-157, `pretrain_step` :455-469): a dense lattice of points inside the edit
  bounds is mapped to source space, the teacher is queried ONCE for (sigma, colour) targets, then the student is fitted
  with L1(sigma) + L1(colour) on point batches, MLPs frozen (`freeze_mlp` :472-488) — pure encoder fwd/bwd, no marching;
* global fine-tuning (`train_step` :589-594 + nerf/utils.py:436-537): ordinary ray batches whose targets (RGB + depth)
  are rendered by the teacher through the proxy (`proxy_truth` :506-586); loss MSE(rgb) + L1(depth).
Sharding: every rank takes a contiguous shard of each point chunk / its own ray batch; gradients are summed
through `parallel.RayShardedDP`'s flat bucket (one RCCL all-reduce per step)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from nerf.trainer import Trainer


def _euler_dirs(angle_step):
    """unit directions of `Rotation.from_euler('xyz', eulers, degrees=True).apply([1-1e-5, 0, 0])` (trainer.py:627-631)"""
    a = np.deg2rad(np.arange(0, 360, angle_step, dtype=np.float64))
    rx, ry, rz = np.meshgrid(a, a, a, indexing="ij")
    rx, ry, rz = rx.reshape(-1), ry.reshape(-1), rz.reshape(-1)
    # extrinsic xyz: R = Rz @ Ry @ Rx ; R @ [s,0,0] = s * first column
    s = 1 - 1e-5
    cy, sy, cz, sz = np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    return torch.from_numpy(np.stack([s * cz * cy, s * sz * cy, -s * sy], -1))


def sample_points(bounds, point_step=0.005, angle_step=45):
    """lattice points inside `bounds` ((2,3) or (B,2,3)) + the Euler-grid direction set — trainer.py:609-635"""
    if bounds.ndim == 2:
        bounds = bounds[None]
    pts, dirs = [], []
    for i in range(bounds.shape[0]):
        lo, hi = bounds[i].cpu()
        X, Y, Z = torch.meshgrid(torch.arange(lo[0], hi[0], step=point_step), torch.arange(lo[1], hi[1], step=point_step),
                                 torch.arange(lo[2], hi[2], step=point_step), indexing="ij")
        pts.append(torch.stack([X, Y, Z], dim=-1).reshape(-1, 3))
        dirs.append(_euler_dirs(angle_step))
    return torch.cat(pts), torch.cat(dirs)


def freeze_module(module, freeze):
    module.training = not freeze
    for p in module.parameters():
        p.requires_grad = not freeze


class SealTrainer(Trainer):
    def __init__(self, student, teacher, lr=1e-2, fp16=True, dist=None, depth_weight=1.0, **kw):
        super().__init__(student, lr=lr, fp16=fp16, dist=dist, **kw)
        self.teacher = teacher
        self.teacher.eval()
        self.depth_weight = depth_weight
        self.pretraining_data = {}
        self.base_lr = lr

    # ------------------------------------------------------------------ local pretraining
    @torch.no_grad()
    def init_pretraining(self, batch_size=6144000, lr=0.05, local_point_step=0.005, local_angle_step=45, seed=0):
        mapper = self.teacher.seal_mapper
        dev = next(self.model.parameters()).device
        pts, dirs = sample_points(mapper.map_data["force_fill_bound"], local_point_step, local_angle_step)
        pts, dirs = pts.to(dev, torch.float32), dirs.to(dev, torch.float32)
        ones = torch.zeros_like(pts) + torch.tensor([1.0, 0, 0], device=dev)
        mapped_p, mapped_d, mask = mapper.map_to_origin(pts, ones)
        if "map_source" in mapper.map_data:
            mask[:] = True
        pts = pts[mask]
        g = torch.Generator(device="cpu").manual_seed(seed)
        dirs = dirs[torch.randint(dirs.shape[0], (pts.shape[0],), generator=g).to(dev)]
        mapped_p, mapped_d = mapped_p[mask], mapped_d[mask]
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16 and dev.type == "cuda"):
            chunks = [self.teacher(mapped_p[i:i + (1 << 20)], mapped_d[i:i + (1 << 20)]) for i in range(0, pts.shape[0], 1 << 20)]
        gt_sigma = torch.cat([c[0].float() for c in chunks])
        gt_color = mapper.map_color(mapped_p, mapped_d, torch.cat([c[1].float() for c in chunks]))
        steps = list(range(0, pts.shape[0], batch_size))
        if steps[-1] != pts.shape[0]:
            steps.append(pts.shape[0])
        self.pretraining_data["local"] = {"points": pts, "dirs": dirs, "sigma": gt_sigma, "color": gt_color, "steps": steps}
        self.pretraining_lr = lr
        return pts.shape[0]

    def freeze_mlp(self, freeze=True):
        for name in ("sigma_net", "color_net", "bg_net"):
            m = getattr(self.model, name, None)
            if m is not None:
                freeze_module(m, freeze)

    def set_lr(self, lr):
        for g in self.optimizer.param_groups:
            g["lr"] = lr

    def pretrain_loss(self, points, dirs, gt_sigma, gt_color, n_total=None):
        """SealNeRF/trainer.py:455-469: L1Loss(sigma) + L1Loss(colour) (means) of the student on one point chunk.  With a
        shard of the chunk, `n_total` is the size of the whole chunk: the shard's sums are normalised by the global count."""
        n_total = n_total or points.shape[0]
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            sigma, color = self.model(points, dirs)
            return (sigma.float() - gt_sigma).abs().sum() / n_total + (color.float() - gt_color).abs().sum() / (n_total * 3)

    def finetune_loss(self, rays_o, rays_d, gt_rgb, gt_depth=None, bg_color=1):
        """nerf/utils.py:436-537 with Seal's depth target: mean over rays of (MSE over channels + L1Loss(depth)) = MSE + L1"""
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            out = self.model.render(rays_o, rays_d, bg_color=bg_color, perturb=True, force_all_rays=False, **self.render_kwargs)
            loss = F.mse_loss(out["image"], gt_rgb)
            if gt_depth is not None:
                loss = loss + self.depth_weight * F.l1_loss(torch.nan_to_num(out["depth"], nan=0.0).view(gt_depth.shape), gt_depth)
        return loss, out

    def pretrain_step(self, points, dirs, gt_sigma, gt_color, n_total=None):
        """one optimizer step on one point shard; `n_total` = size of the un-sharded chunk (mean over all ranks)"""
        self.model.train()
        self.optimizer.zero_grad(set_to_none=False)
        world = self.dist.world if self.dist is not None else 1
        # x world because the DP layer averages the shards' gradients
        loss = self.pretrain_loss(points, dirs, gt_sigma, gt_color, n_total) * world
        self.scaler.scale(loss).backward()
        if self.dist is not None:
            self.dist.allreduce_grads(self.scaler)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        return loss.detach() / world

    def pretrain_one_epoch(self):
        """one pass over the local points (trainer.py:363-452); every rank processes its shard of each chunk"""
        from parallel import shard_slice
        if not self.model.density_bitfield_hacked:
            self.model.hack_bitfield()
        self.set_lr(self.pretraining_lr)
        self.freeze_mlp(True)
        src = self.pretraining_data["local"]
        rank, world = (self.dist.rank, self.dist.world) if self.dist is not None else (0, 1)
        total, n = 0.0, 0
        for a, b in zip(src["steps"][:-1], src["steps"][1:]):
            lo, hi = shard_slice(b - a, rank, world)
            sl = slice(a + lo, a + hi)
            total = total + self.pretrain_step(src["points"][sl], src["dirs"][sl], src["sigma"][sl], src["color"][sl], n_total=b - a)
            n += 1
        self.freeze_mlp(False)
        self.set_lr(self.base_lr)
        return total / max(n, 1)

    # ------------------------------------------------------------------ global fine-tuning
    @torch.no_grad()
    def proxy_truth(self, rays_o, rays_d):
        """teacher-rendered RGB + depth targets for a ray batch (force_all_rays, no perturbation) — trainer.py:506-586"""
        if not self.teacher.density_bitfield_hacked:
            self.teacher.hack_bitfield()
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            out = self.teacher.render(rays_o, rays_d, bg_color=None, perturb=False, force_all_rays=True, **self.render_kwargs)
        return torch.nan_to_num(out["image"], nan=0.0), torch.nan_to_num(out["depth"], nan=0.0)

    def train_step(self, rays_o, rays_d, gt_rgb=None, gt_depth=None, bg_color=1):
        if gt_rgb is None:
            gt_rgb, gt_depth = self.proxy_truth(rays_o, rays_d)
        model = self.model
        model.train()
        self._maybe_update_extra_state()  # (with data parallelism: occupancy state re-synchronised over the ranks)
        self.global_step += 1
        self.optimizer.zero_grad(set_to_none=False)
        loss, _ = self.finetune_loss(rays_o, rays_d, gt_rgb, gt_depth, bg_color)
        self.scaler.scale(loss).backward()
        if self.dist is not None:
            self.dist.allreduce_grads(self.scaler)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        return loss.detach()
