"""Grid-encoder backward: time the table-gradient paths (1 = atomics, 2 = binned) on uniform and ray-coherent (training-like) points.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split of the binned path."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip  # noqa: E402
from nerf import synthetic as syn  # noqa: E402
from tools.microbench import grid_meta, timeit  # noqa: E402

G, R = s3d_hip.GridBackend, s3d_hip.RaymarchingBackend
NAMES = ("", "atomics", "binned")


def coherent_points(n_rays=4096, dev="cuda"):
    grid, bits = syn.lego_like_density_grid(seed=0)
    bits = torch.from_numpy(bits).to(dev)
    poses = syn.orbit_poses(1, seed=0)
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=n_rays, generator=torch.Generator().manual_seed(0))
    ro, rd = r["rays_o"][0].contiguous().to(dev), r["rays_d"][0].contiguous().to(dev)
    N = ro.shape[0]
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    R.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
    M = N * 160
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    R.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                       torch.rand(N, device=dev))
    m = int(counter[0])
    return ((xyzs[:m] + 1) / 2).contiguous()


def main():
    paths = [int(p) for p in (sys.argv[1] if len(sys.argv) > 1 else "2").split(",")]
    only = sys.argv[2] if len(sys.argv) > 2 else ""      # e.g. "uniform262144" / "rays4096"
    dts = {"f16": (torch.float16,), "f32": (torch.float32,)}.get(sys.argv[3] if len(sys.argv) > 3 else "", (torch.float16, torch.float32))
    offs, S, total = grid_meta()
    cases = [("uniform", torch.rand(1 << 15, 3, device="cuda")), ("uniform", torch.rand(1 << 18, 3, device="cuda")),
             ("uniform", torch.rand(1 << 20, 3, device="cuda")), ("rays4096", coherent_points(4096)),
             ("rays16384", coherent_points(16384))]
    r = coherent_points(4096)
    cases.append(("rays4096s", r[torch.randperm(r.shape[0], device="cuda")].contiguous()))  # same points, order shuffled
    cases = [(n, x) for n, x in cases if not only or only == (n + (str(x.shape[0]) if n == 'uniform' else ''))]
    for dtype in dts:
        emb = torch.zeros(total, 2, device="cuda", dtype=dtype)
        for name, x in cases:
            B = x.shape[0]
            grad = (torch.randn(16, B, 2, device="cuda") * 1e-3).to(dtype)
            if name.startswith("rays"):
                grad[:, torch.rand(B, device="cuda") < 0.35] = 0  # samples behind early termination carry no gradient
            ref = None
            for path in paths:
                G.set_backward_path(path)
                ge = torch.zeros(total, 2, device="cuda", dtype=dtype)
                G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0)
                same = "" if ref is None else f" identical_to_first={bool(torch.equal(ref, ge))}"
                ref = ge.clone() if ref is None else ref
                t = timeit(lambda: G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0), iters=10)
                print(f"grid_bwd[{NAMES[path]:9s}] {str(dtype):14s} {name:9s} B={B:8d}: {t*1e6:9.1f} us  {B/t/1e9:7.3f} Gpts/s{same}", flush=True)
            G.set_backward_path(0)


if __name__ == "__main__":
    main()
