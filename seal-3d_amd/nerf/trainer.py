"""Minimal NGP training step (the hot-path part of nerf/utils.py `Trainer`): rays -> render -> MSE -> backward ->
Adam, with fp16 autocast + GradScaler (`-O`), `update_extra_state` every 16 steps (nerf/utils.py:845-848),
Adam(betas=(0.9, 0.99), eps=1e-15) and lr 1e-2 (main_SealNeRF.py:283-288), PSNR as in nerf/utils.py:226-233.
Optional data parallelism over rays: gradients are all-reduced through one flat bucket (parallel/dist.py)."""
import math

import torch
import torch.nn.functional as F


def psnr(pred, target):
    return -10 * math.log10(float(torch.mean((pred.float() - target.float()) ** 2)) + 1e-20)


class Trainer:
    def __init__(self, model, lr=1e-2, fp16=True, update_extra_interval=16, dist=None, max_steps=1024, dt_gamma=0,
                 T_thresh=1e-4):
        self.model = model
        self.fp16 = fp16
        self.update_extra_interval = update_extra_interval
        self.dist = dist
        self.render_kwargs = dict(max_steps=max_steps, dt_gamma=dt_gamma, T_thresh=T_thresh)
        on_gpu = next(model.parameters()).is_cuda
        self.optimizer = torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15, fused=on_gpu)
        self.scaler = torch.amp.GradScaler("cuda", enabled=fp16)
        self.global_step = 0
        if dist is not None:
            dist.register(model)

    def train_step(self, rays_o, rays_d, gt_rgb, bg_color=1):
        """rays_o/d [N,3], gt_rgb [N,3].  Returns the (detached) loss tensor; no host sync."""
        model = self.model
        model.train()
        if model.cuda_ray and self.global_step % self.update_extra_interval == 0:
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
                model.update_extra_state()
        self.global_step += 1
        self.optimizer.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            out = model.render(rays_o, rays_d, bg_color=bg_color, perturb=True, force_all_rays=False, **self.render_kwargs)
            loss = F.mse_loss(out["image"], gt_rgb)
        self.scaler.scale(loss).backward()
        if self.dist is not None:
            self.dist.allreduce_grads(self.scaler)
        self.scaler.step(self.optimizer)
        self.scaler.update()
        return loss.detach()

    @torch.no_grad()
    def render_image(self, rays_o, rays_d, bg_color=1):
        self.model.eval()
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            return self.model.render(rays_o, rays_d, bg_color=bg_color, perturb=False, **self.render_kwargs)
