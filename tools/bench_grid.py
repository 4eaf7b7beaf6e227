#!/usr/bin/env python3
"""Isolated hash-grid encoder benchmark (GPU box): forward and backward of the Lego configuration (L16 F2 T2^19, fp16
tables) on random points and on ray-ordered points (samples marched through the synthetic scene), B = 2^18 and 2^21.
Prints one line per (kernel, order, B): time per launch (HIP events), points/s and the fraction of the 8 TB/s HBM roofline
at SURVEY §8(d)'s 588 algorithmic bytes per point.  `--sum` prints checksums of the outputs (variants must agree bit for
bit); `--iters` bounds the launches (rocprofv3 --pmc passes)."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "seal-3d_amd")):
    sys.path.insert(0, p)
import s3d_hip  # noqa: E402
from nerf import synthetic as syn  # noqa: E402

BYTES_PT = 588


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


_FLUSH = None


def timeit_cold(fn, iters, mbytes=1024):
    """per-launch HIP events with a cache-thrashing pass (read-modify-write of `mbytes` MiB) before every launch: the time
    of a launch that finds neither the table nor the points in L2 / MALL, as inside a training step"""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.zeros(mbytes << 18, dtype=torch.float32, device="cuda")
    tot = 0.0
    for _ in range(iters):
        _FLUSH.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / iters * 1e-3


def grid_meta(dev):
    pls = np.exp2(np.log2(2048 / 16) / 15)
    offs, off = [], 0
    for i in range(16):
        res = int(np.ceil(16 * pls ** i))
        n = int(np.ceil(min(2 ** 19, (res + 1) ** 3) / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return torch.tensor(offs, dtype=torch.int32, device=dev), float(np.log2(pls)), off


def ray_ordered_points(B, dev):
    """xyz in [0,1] of samples marched along 800x800 Lego-camera rays through the synthetic occupancy (training order)"""
    R = s3d_hip.RaymarchingBackend
    _, bits = syn.lego_like_density_grid(seed=0)
    bits = torch.from_numpy(bits).to(dev)
    out, have, k = [], 0, 0
    while have < B:
        poses = syn.orbit_poses(8, seed=k)
        r = syn.get_rays(poses[k % 8:k % 8 + 1], syn.lego_intrinsics(), 800, 800, N=16384, generator=torch.Generator().manual_seed(k))
        ro, rd = r["rays_o"][0].contiguous().to(dev), r["rays_d"][0].contiguous().to(dev)
        N = ro.shape[0]
        aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
        nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
        R.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
        M = N * 128
        xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        R.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                           torch.rand(N, device=dev))
        m = min(int(counter[0]), M)
        out.append(((xyzs[:m] + 1) / 2))
        have += m
        k += 1
    return torch.cat(out)[:B].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sizes", type=int, nargs="+", default=[1 << 18, 1 << 21])
    ap.add_argument("--orders", nargs="+", default=["random", "ray"])
    ap.add_argument("--no_bwd", action="store_true")
    ap.add_argument("--no_fwd", action="store_true")
    ap.add_argument("--sum", action="store_true")
    ap.add_argument("--cold", action="store_true", help="thrash L2/MALL before every timed launch")
    ap.add_argument("--bwd_path", type=int, default=0, help="`path` of s3d_grid_encode_backward (0 auto, 1 atomics, 2 binned)")
    a = ap.parse_args()
    if a.cold:
        global timeit
        timeit = lambda fn, iters, warm=2: timeit_cold(fn, iters)  # noqa: E731
    dev = "cuda"
    torch.manual_seed(0)
    G = s3d_hip.GridBackend
    G.set_backward_path(a.bwd_path)
    offs, S, total = grid_meta(dev)
    emb = (torch.rand(total, 2, device=dev) * 2 - 1).half()
    print("device:", torch.cuda.get_device_name(0), " S3D_GRID_FWD=", os.environ.get("S3D_GRID_FWD", ""), " S3D_GRID_BWD=",
          os.environ.get("S3D_GRID_BWD", ""))
    for B in a.sizes:
        for order in a.orders:
            x = torch.rand(B, 3, device=dev) if order == "random" else ray_ordered_points(B, dev)
            out = torch.empty(16, B, 2, device=dev, dtype=torch.half)
            if not a.no_fwd:
                t = timeit(lambda: G.grid_encode_forward(x, emb, offs, out, B, 3, 2, 16, S, 16, None, 0, False, 0), a.iters)
                print(f"grid_fwd f16 {order:6s} B={B:8d}: {t*1e6:9.1f} us  {B/t/1e9:6.2f} Gpts/s  frac_hbm={B*BYTES_PT/t/8e12:.3f}"
                      + (f"  sum={out.float().double().sum().item():.10e} absum={out.float().abs().double().sum().item():.10e}" if a.sum else ""))
            if not a.no_bwd:
                grad = (torch.randn(16, B, 2, device=dev) * 1e-3).half()
                if order == "ray":  # samples behind a ray's termination carry exact-zero gradients in training: the last
                    # fifth of every 64-sample stretch (a ray of the Lego step has ~64 samples), not scattered singles
                    grad[:, (torch.arange(B, device=dev) % 64) >= 51] = 0
                ge = torch.zeros(total, 2, device=dev, dtype=torch.half)

                def bwd():
                    G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0)
                t = timeit(bwd, max(a.iters // 2, 1))
                ge.zero_()
                bwd()
                print(f"grid_bwd f16 {order:6s} B={B:8d}: {t*1e6:9.1f} us  {B/t/1e9:6.2f} Gpts/s  frac_hbm={B*BYTES_PT/t/8e12:.3f}"
                      + (f"  sum={ge.float().double().sum().item():.10e} absum={ge.float().abs().double().sum().item():.10e}" if a.sum else ""))


if __name__ == "__main__":
    main()
