#!/usr/bin/env python3
"""Phase times of k_march_count_wave (library built with -DS3D_MARCH_PROFILE: tools/build_variants.sh raymarching
"mprof:-DS3D_MARCH_PROFILE", run with S3D_HIP_LIB=<that .so>): A = t sequence of a window, B = probes + skip targets,
C = pointer doubling, D = emission + carry.  100 MHz ticks per ray, training-like rays of the bench scene."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip  # noqa: E402
from nerf import synthetic as syn  # noqa: E402

R = s3d_hip.RaymarchingBackend
dev = "cuda"
N, max_steps = 4096, 1024
grid, bits = syn.lego_like_density_grid(seed=0)
bits = torch.from_numpy(bits).to(dev)
poses = syn.orbit_poses(1, seed=0)
r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=N, generator=torch.Generator().manual_seed(0))
ro, rd = r["rays_o"][0].contiguous().to(dev), r["rays_d"][0].contiguous().to(dev)
aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
R.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
M = N * 160
xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
noises = torch.rand(N, device=dev)
for _ in range(3):
    counter.zero_()
    R.march_rays_train(ro, rd, bits, 1.0, 0.0, max_steps, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
torch.cuda.synchronize()
nbytes = s3d_hip.lib().s3d_march_rays_train_workspace_size(N, max_steps)
ws = s3d_hip._ws.get(nbytes, ro.device)
f = ws[:(4 + N + N * max_steps) * 4].view(torch.float32)  # u32 header[4] | counts[N] | float tsamples[N * max_steps]
rows = f[4 + N:4 + N + N * max_steps].view(N, max_steps)
tp = rows[:, max_steps - 8:max_steps - 4].cpu().numpy().astype(np.float64)
cnt = rays[:, 2].cpu().numpy()
print(f"{N} rays, {int(cnt.sum())} samples ({cnt.mean():.1f} per ray, max {cnt.max()})")
for i, name in enumerate(("A t-sequence", "B probes", "C doubling", "D emit+carry")):
    print(f"  {name:14s}: mean {tp[:, i].mean() * 10:8.1f} ns   p50 {np.percentile(tp[:, i], 50) * 10:8.1f}   max {tp[:, i].max() * 10:8.1f}")
print(f"  rays whose phase C took the pointer-doubling rounds (> 5 us): {int((tp[:, 2] * 10 > 5000).sum())} of {N}")
dg = rows[:, max_steps - 16:max_steps - 12].cpu().numpy()
bad = np.nonzero(dg[:, 0] > 0)[0]
print(f"  rays whose candidate set differed from the accepted marks: {len(bad)}")
for n_ in bad[:24]:
    print(f"    ray {n_:5d}: {int(dg[n_, 0]):4d} entries differ, first at {int(dg[n_, 1]):4d}, window {int(dg[n_, 2]):4d}, start {int(dg[n_, 3]):4d}, samples {int(cnt[n_])}, "
          f"d = ({rd[n_, 0].item():+.4f} {rd[n_, 1].item():+.4f} {rd[n_, 2].item():+.4f})")
tot = tp.sum(1)
print(f"  total         : mean {tot.mean() * 10:8.1f} ns   p50 {np.percentile(tot, 50) * 10:8.1f}   p99 {np.percentile(tot, 99) * 10:8.1f}   max {tot.max() * 10:8.1f}")
