import os, sys, torch, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
from nerf import synthetic as syn
R = s3d_hip.RaymarchingBackend
dev = "cuda"
grid, bits = syn.lego_like_density_grid(seed=0)
bits = torch.from_numpy(bits).to(dev)
poses = syn.orbit_poses(1, seed=0)
r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=4096, generator=torch.Generator().manual_seed(0))
ro, rd = r["rays_o"][0].contiguous().to(dev), r["rays_d"][0].contiguous().to(dev)
N = ro.shape[0]
aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
R.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
M = N * 160
xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
for _ in range(3):
    counter.zero_()
    R.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, torch.rand(N, device=dev))
torch.cuda.synchronize()
ws = list(s3d_hip._ws.buf.values())[0]
t = ws[(4 + N) * 4: (4 + N) * 4 + N * 1024 * 4].view(torch.float32).view(N, 1024)[:, 1016:1020].cpu().numpy()
print("samples", int(counter[0]), "mean ticks (10 ns) per ray  A,B,C,emit:", t.mean(0), "max:", t.max(0), "sum mean us:", t.sum(1).mean() / 100)
steps = ((fars - nears) / (2 * 3 ** 0.5 / 1024)).cpu().numpy()
print("mean window length", steps.mean(), "max", steps.max())
