"""CPU, gloo, 4 ranks (+ the same code on a 1-rank group): the data-parallel pieces of SURVEY §8(e) that need no GPU —
the occupancy sweep split over the ranks (density queries all-gathered, a common random stream: identical grids on every
rank, no broadcast), a frame rendered in contiguous pixel ranges (image + depth all-gathered), the gradient reduction
issued in pieces — give bit for bit what one rank computes alone."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO

_SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
repo = os.environ["S3D_REPO"]
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "seal-3d_amd"))
from parallel import RayShardedDP, init_from_env, shard_slice
rank, world, _ = init_from_env("gloo")
if world == 1:
    dist.init_process_group(backend="gloo", rank=0, world_size=1)
from oracle import oracle_backend as ob
import raymarching.raymarching as rm
rm._backend = ob.RaymarchingBackend
ob.set_threads(1)
from nerf import renderer, synthetic as syn
lo, hi = syn.lego_like_boxes(0)

class Analytic(renderer.NeRFRenderer):
    def __init__(self, **kw):
        super().__init__(**kw)
        self.w = torch.nn.Parameter(torch.ones(3))
    def forward(self, x, dd):
        return syn.box_density(x, lo, hi, sigma=40.0), (x * 0.5 + 0.5).clamp(0, 1) * (0.5 + 0.5 * dd.abs()) * self.w
    def density(self, x):
        return {"sigma": syn.box_density(x, lo, hi, sigma=40.0)}

torch.manual_seed(100 + rank)   # the ranks' own generators differ: the sweep must not depend on them
R = Analytic(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
R.grid_size = 32
R.density_grid = torch.zeros(1, 32 ** 3)
R.density_bitfield = torch.zeros(32 ** 3 // 8, dtype=torch.uint8)
dp = RayShardedDP(force_collective=True).register(R)
assert R.dist_shard is dp
R.train()
for it in range(18):            # 16 full sweeps, then two partial updates (occupied-cell picks)
    R.iter_density = max(R.iter_density, 15 if it == 1 else R.iter_density)  # (skip most of the full sweeps)
    R.update_extra_state()
    dp.sync_extra_state(R)
grids = [torch.empty_like(R.density_grid) for _ in range(world)]
dist.all_gather(grids, R.density_grid)
same = all(torch.equal(grids[0], g) for g in grids)
# a frame in pixel ranges
poses = syn.orbit_poses(1, seed=0)
r = syn.get_rays(poses, syn.lego_intrinsics(48, 48), 48, 48)
R.eval()
R.device_compaction = False
with torch.no_grad():
    out = dp.sharded_render(R, r["rays_o"], r["rays_d"], bg_color=1, perturb=False, max_steps=1024)
# gradient pieces: a fake fp16-style buffer with parameter cuts + the fp32 bucket
buf = torch.arange(10000, dtype=torch.float32) * (rank + 1)
buf._s3d_param_cuts = [0, 4000, 4096]
dp.half_grads = [buf]
dp.chunk_bytes = 4096
pieces = dp.grad_chunks()
cover = sorted((a, b) for h, a, b in pieces if h is buf)
contig = cover[0][0] == 0 and cover[-1][1] == 10000 and all(x[1] == y[0] for x, y in zip(cover, cover[1:]))
dp.allreduce_grads()
mean = torch.arange(10000, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
red_ok = torch.allclose(buf, mean, rtol=1e-6)
if rank == 0:
    torch.save({"grid": R.density_grid, "bits": R.density_bitfield, "image": out["image"], "depth": out["depth"],
                "mean_density": R.mean_density}, os.environ["S3D_OUT"])
print(f"RANK{rank} same={same} contig={contig} npieces={len(cover)} red_ok={red_ok}")
dist.destroy_process_group()
'''


def _run(tmp_path, nproc, port):
    script = tmp_path / "shard.py"
    script.write_text(_SCRIPT)
    out = tmp_path / f"out{nproc}.pt"
    env = dict(os.environ, S3D_REPO=REPO, S3D_OUT=str(out), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    if nproc == 1:
        cmd = [sys.executable, str(script)]
        env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    for r in range(nproc):
        assert f"RANK{r} same=True contig=True" in res.stdout and "red_ok=True" in res.stdout, res.stdout + res.stderr[-1500:]
    return torch.load(out)


def test_sharded_occupancy_sweep_render_and_chunked_reduce_world4_equals_world1(tmp_path):
    a = _run(tmp_path, 4, 29541)
    b = _run(tmp_path, 1, 29542)
    assert torch.equal(a["grid"], b["grid"]) and torch.equal(a["bits"], b["bits"]) and a["mean_density"] == b["mean_density"]
    assert float((a["grid"] > 0).float().mean()) > 0.01
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"])
    assert a["image"].shape == (1, 48 * 48, 3) and float(a["image"].std()) > 0
