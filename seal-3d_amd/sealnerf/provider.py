"""The Seal data provider's distillation half (SealNeRF/provider.py:13-136): a pose set whose images are REPLACED by the
teacher's renders through the proxy (`proxy_dataset`), and the per-step batch that gathers colour + depth targets of random
pixels from them (`collate`).  The disk loader under it (nerf/provider.py: transforms.json, image decoding) is outside the
hot path — poses, intrinsics and the frame size are handed in."""
import torch

from nerf.synthetic import get_rays


class SealDataset:
    def __init__(self, poses, intrinsics, H, W, num_rays=4096, images=None, device=None, training=True, render_kwargs=None,
                 fp16=False):
        self.device = torch.device(device) if device is not None else poses.device
        self.poses = poses.to(self.device)
        self.intrinsics, self.H, self.W = intrinsics, H, W
        self.training = training
        self.num_rays = num_rays if training else -1
        self.images = images  # [B, H, W, 3] or None (the reference's ground truth: only its shape is used by proxy_dataset)
        self.depths = None
        self.proxy_flag = False
        self.fp16 = fp16
        self.render_kwargs = dict(render_kwargs or {})

    def __len__(self):
        return self.poses.shape[0]

    @torch.no_grad()
    def proxy_dataset(self, model, n_batch=1):
        """SealNeRF/provider.py:19-70: every pose rendered by the teacher (`render(..., staged=True, bg_color=None,
        perturb=False, force_all_rays=True)` in the mode the teacher is in, in `n_batch` pieces — like the reference's loop,
        a remainder piece makes `n_batch` one larger FOR EVERY LATER POSE too), NaNs zeroed; the frames become `images`
        [B, H, W, 3] and `depths` [B, H, W, 1] and batches are marked `skip_proxy`."""
        images, depths = [], []
        if not getattr(model, "density_bitfield_hacked", True):
            model.hack_bitfield()  # (SealNeRF/trainer.py:268-269, before the provider is asked)
        for i in range(len(self)):
            rays = get_rays(self.poses[i:i + 1], self.intrinsics, self.H, self.W, -1)
            rays_o, rays_d = rays["rays_o"].contiguous(), rays["rays_d"].contiguous()
            total = rays_o.shape[1]
            size = total // n_batch
            if total % n_batch:
                n_batch += 1
            img, dep = [], []
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16 and rays_o.is_cuda):
                for k in range(n_batch):
                    out = model.render(rays_o[:, k * size:(k + 1) * size], rays_d[:, k * size:(k + 1) * size], staged=True,
                                       bg_color=None, perturb=False, force_all_rays=True, **self.render_kwargs)
                    img.append(out["image"])
                    dep.append(out["depth"])
            img = torch.nan_to_num(torch.cat(img, 1), nan=0.0)
            dep = torch.nan_to_num(torch.cat(dep, 1), nan=0.0)
            images.append(img.float().view(self.H, self.W, -1))
            depths.append(dep.float().view(self.H, self.W, -1))
        self.images = torch.stack(images, dim=0)
        self.depths = torch.stack(depths, dim=0)
        self.proxy_flag = True

    def collate(self, index, generator=None):
        """SealNeRF/provider.py:72-128 for a dataset pose: `num_rays` random pixels of pose `index[0]`, their rays and the
        targets gathered from `images` / `depths`"""
        B = len(index)
        poses = self.poses[index]
        rays = get_rays(poses, self.intrinsics, self.H, self.W, self.num_rays, generator=generator)
        out = {"H": self.H, "W": self.W, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "skip_proxy": self.proxy_flag,
               "data_index": torch.tensor(index), "pixel_index": rays["inds"] if self.num_rays > 0 else None}
        for name, frames in (("images", self.images), ("depths", self.depths)):
            if frames is None:
                continue
            v = frames[index].to(self.device)
            if self.training:
                C = v.shape[-1]
                v = torch.gather(v.view(B, -1, C), 1, torch.stack(C * [rays["inds"]], -1))
            out[name] = v
        return out
