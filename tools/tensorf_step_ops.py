#!/usr/bin/env python3
"""Every device kernel of ONE TensoRF VM-48 training step (resolution 300, native trainer), in launch order, with the aten op
that launched it: the list the step's launch-bound part is read from (torch.profiler; run on the GPU box)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "seal-3d_amd")):
    sys.path.insert(0, p)
from nerf import synthetic as syn  # noqa: E402
from tensoRF import network as trf  # noqa: E402
from tensoRF.utils import Trainer  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    poses = syn.orbit_poses(8, seed=0).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    torch.manual_seed(0)
    net = trf.NeRFNetwork(resolution=[res] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    net.density_grid.copy_(torch.from_numpy(grid))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    net.iter_density = 100
    tr = Trainer(net, lr0=2e-2, lr1=1e-3, fp16=True, update_extra_interval=10 ** 9, native_optim=True)
    tr.global_step = 1
    batches = []
    for k in range(8):
        r = syn.get_rays(poses[k:k + 1], syn.lego_intrinsics(), 800, 800, N=4096, generator=torch.Generator().manual_seed(k))
        batches.append((r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous(), torch.rand(4096, 3, device="cuda")))
    for k in range(4):
        tr.train_step(*batches[k % 8])
    net.mean_count = int(net.step_counter[:4, 0].float().mean().item())
    net.local_step = 0
    for k in range(6):
        tr.train_step(*batches[k % 8])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        tr.train_step(*batches[0])
        torch.cuda.synchronize()
    rows = []
    for ev in sorted(prof.events(), key=lambda e: e.time_range.start):
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.kernels and not any(c.kernels for c in ev.cpu_children):
            for k in ev.kernels:  # (innermost op that launched something)
                rows.append((k.duration, k.name[:60], ev.name[:40], str(ev.input_shapes)[:70]))
    seen, total = rows, 0.0
    print(f"{'us':>7}  kernel | op | shapes")
    for d, k, op, sh in rows:
        total += d
        print(f"{d:7.1f}  {k} | {op} | {sh}")
    print(f"{len(seen)} kernels, {total:.0f} us of kernel time")


if __name__ == "__main__":
    main()
