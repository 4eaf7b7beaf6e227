"""Grid backward + Adam on ray-ordered marched samples: separate (s3d_grid_encode_backward + s3d_adam_step_multi) vs the update
inside the accumulate kernel (s3d_grid_encode_backward_adam).  S3D_HIP_LIB selects a library variant (tools/build_variants.sh)."""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
import raymarching
from gridencoder import GridEncoder
from nerf import synthetic as syn
from tools.microbench import timeit
G, O = s3d_hip.GridBackend, s3d_hip.OptimBackend
enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)
offs = enc.offsets.cuda(); rows = int(offs[-1]); L = 16; S = float(np.log2(enc.per_level_scale))
grid, bits = syn.lego_like_density_grid(seed=0)
poses = syn.orbit_poses(2, seed=0).cuda()
r = syn.get_rays(poses[:1], syn.lego_intrinsics(), 800, 800, N=int(os.environ.get("S3D_RAYS", 4096)), generator=torch.Generator().manual_seed(0))
ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device="cuda")
nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
counter = torch.zeros(2, dtype=torch.int32, device="cuda")
xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, torch.from_numpy(bits).cuda(), 1, 128, nears, fars, counter, 0, False, 128, True, 0, 1024)
B = (int(counter[0]) + 127) // 128 * 128
x = ((xyzs[:B] + 1) / 2).contiguous()
grad = (torch.randn(L, B, 2, device="cuda") * 3).half()
p = (torch.rand(rows, 2, device="cuda") * 2e-4 - 1e-4); m = torch.zeros_like(p); v = torch.zeros_like(p); h = p.half()
ge = torch.zeros(rows, 2, dtype=torch.half, device="cuda")
step = torch.zeros(1, device="cuda"); scale = torch.full((1,), 1024.0, device="cuda"); flag = torch.zeros(1, device="cuda")
adam = dict(param=p, exp_avg=m, exp_avg_sq=v, param_half=h, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, step=step, grad_scale=scale, lr_scale=None)
t_b = timeit(lambda: G.grid_encode_backward(grad, x, h, offs, ge, B, 3, 2, L, S, 16, None, None, 0, False, 0, found_inf=flag), iters=20)
t_a = timeit(lambda: O.adam_step_multi([(p, ge, m, v, h, 1e-2, 0.9, 0.99, 1e-15, True)], step, scale, flag), iters=20)
def both():
    G.grid_encode_backward(grad, x, h, offs, ge, B, 3, 2, L, S, 16, None, None, 0, False, 0, found_inf=flag)
    O.adam_step_multi([(p, ge, m, v, h, 1e-2, 0.9, 0.99, 1e-15, True)], step, scale, flag)
t_ba = timeit(both, iters=20)
t_f = timeit(lambda: G.grid_encode_backward_adam(grad, x, h, offs, ge, B, 3, 2, L, S, 16, 0, False, 0, adam, found_inf=flag), iters=20)
alg = 588 * B + 30 * rows * 2
print(f"B={B} rows={rows}: backward {t_b*1e6:.1f} us + adam {t_a*1e6:.1f} us = {t_ba*1e6:.1f} us back to back | fused {t_f*1e6:.1f} us "
      f"({alg / t_f / 1e12:.2f} TB/s of 588 B x points + 30 B x parameters)", flush=True)
