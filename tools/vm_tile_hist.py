"""Points per occupied 8x8 plane tile / 64-cell line chunk of one 4,096-ray batch of the synthetic Lego scene at resolution 300
(what the TensoRF factor backward's segments look like)."""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip  # noqa
import raymarching
from nerf import synthetic as syn
grid, bits = syn.lego_like_density_grid(seed=0)
poses = syn.orbit_poses(2, seed=0).cuda()
r = syn.get_rays(poses[:1], syn.lego_intrinsics(), 800, 800, N=4096, generator=torch.Generator().manual_seed(0))
ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device="cuda")
nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
counter = torch.zeros(2, dtype=torch.int32, device="cuda")
xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, torch.from_numpy(bits).cuda(), 1, 128, nears, fars, counter, 0, False, 128, True, 0, 1024)
B = int(counter[0]); x = xyzs[:B]
res = 300
print("points", B)
for (u, v) in ((0, 1), (0, 2), (1, 2)):
    ix = ((x[:, u] + 1) / 2 * (res - 1)).floor().clamp(0, res - 1).long() // 8
    iy = ((x[:, v] + 1) / 2 * (res - 1)).floor().clamp(0, res - 1).long() // 8
    key = iy * 38 + ix
    cnt = torch.bincount(key, minlength=38 * 38).cpu().numpy()
    occ = cnt[cnt > 0]
    qs = np.percentile(occ, [10, 25, 50, 75, 90, 99])
    print(f"plane ({u},{v}): occupied tiles {len(occ)} of {38*38}; points/tile mean {occ.mean():.0f} quantiles 10/25/50/75/90/99 = {qs}; "
          f"tiles < 16 pts: {(occ < 16).sum()} holding {occ[occ < 16].sum()} pts; < 64: {(occ < 64).sum()} holding {occ[occ < 64].sum()}; > 512: {(occ > 512).sum()} holding {occ[occ > 512].sum()}")
    # key changes along the sorted order = segments if one workgroup took everything
    cell = (((x[:, v] + 1) / 2 * (res - 1)).floor().long() * res + ((x[:, u] + 1) / 2 * (res - 1)).floor().long())
    same = (cell[1:] == cell[:-1]).float().mean().item()
    print(f"    consecutive samples in the same plane cell: {same:.2f}; distinct cells {cell.unique().numel()}")
