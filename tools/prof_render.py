"""Render-loop anatomy on the GPU box: iterations, alive counts and per-stage time of one 800x800 frame."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip, raymarching
from nerf import network_ff, synthetic as syn
from nerf.trainer import Trainer
torch.manual_seed(0)
dev = "cuda"
model = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
grid, bits = syn.lego_like_density_grid(seed=0)
model.density_grid.copy_(torch.from_numpy(grid)); model.density_bitfield.copy_(torch.from_numpy(bits))
model.eval()
poses = syn.orbit_poses(1, seed=0).to(dev)
r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800)
ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
N = ro.shape[0]
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for rep in range(2):
        nears, fars = raymarching.near_far_from_aabb(ro, rd, model.aabb_infer, 0.2)
        ws = torch.zeros(N, device=dev); dp = torch.zeros(N, device=dev); im = torch.zeros(N, 3, device=dev)
        alive = torch.arange(N, dtype=torch.int32, device=dev); rays_t = nears.clone()
        n_alive, step, log = N, 0, []
        t0 = time.perf_counter()
        while step < 1024 and n_alive > 0:
            n_step = max(min(4 * N // n_alive, 32), 1)  # infer_batch_scale = 4 (bench default)
            e0 = ev()
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, alive, rays_t, ro, rd, 1.0, model.density_bitfield, 1, 128, nears, fars, 128, False, 0, 1024)
            e1 = ev()
            sig, rgb = model(xyzs, dirs)
            e2 = ev()
            raymarching.composite_rays(n_alive, n_step, alive, rays_t, sig, rgb, deltas, ws, dp, im, 1e-4)
            e3 = ev()
            alive, cnt = raymarching.compact_rays_alive(alive, n_alive)
            n_new = int(cnt.item()); alive = alive[:n_new]
            e4 = ev()
            log.append((n_alive, n_step, e0, e1, e2, e3, e4))
            n_alive = n_new; step += n_step
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
    print(f"frame wall {wall*1e3:.1f} ms, {len(log)} iterations")
    tm = [0, 0, 0, 0]
    for i, (na, ns, e0, e1, e2, e3, e4) in enumerate(log):
        d = [e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e3.elapsed_time(e4)]
        for k in range(4): tm[k] += d[k]
        if i < 12 or i % 10 == 0: print(f"it {i:3d} alive {na:7d} n_step {ns} march {d[0]*1e3:7.0f}us net {d[1]*1e3:7.0f}us comp {d[2]*1e3:6.0f}us compact {d[3]*1e3:6.0f}us")
    print("totals ms: march %.2f net %.2f composite %.2f compact %.2f" % tuple(tm))
