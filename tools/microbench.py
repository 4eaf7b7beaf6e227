#!/usr/bin/env python3
"""Per-kernel micro-benchmarks (GPU box): times each hot kernel with HIP events and prints algorithmic GB/s."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "seal-3d_amd")):
    sys.path.insert(0, p)
import s3d_hip  # noqa: E402
from nerf import synthetic as syn  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


def grid_meta():
    pls = np.exp2(np.log2(2048 / 16) / 15)
    offs, off = [], 0
    for i in range(16):
        res = int(np.ceil(16 * pls ** i))
        n = int(np.ceil(min(2 ** 19, (res + 1) ** 3) / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return torch.tensor(offs, dtype=torch.int32, device="cuda"), float(np.log2(pls)), off


def main():
    dev = "cuda"
    offs, S, total = grid_meta()
    G, R, F = s3d_hip.GridBackend, s3d_hip.RaymarchingBackend, s3d_hip.FFMLPBackend
    print("device:", torch.cuda.get_device_name(0))
    for dtype in (torch.float16, torch.float32):
        s = 2 if dtype == torch.float16 else 4
        emb = (torch.rand(total, 2, device=dev) * 2e-4 - 1e-4).to(dtype)
        for B in (1 << 15, 1 << 17, 1 << 18, 1 << 21):
            x = torch.rand(B, 3, device=dev)
            out = torch.empty(16, B, 2, device=dev, dtype=dtype)
            t = timeit(lambda: G.grid_encode_forward(x, emb, offs, out, B, 3, 2, 16, S, 16, None, 0, False, 0))
            bytes_pt = 12 + 16 * 8 * 2 * s + 16 * 2 * s
            print(f"grid_fwd {dtype} B={B}: {t*1e6:9.1f} us  {B/t/1e9:7.3f} Gpts/s  {B*bytes_pt/t/1e9:8.1f} GB/s algorithmic ({B*bytes_pt/t/8e12*100:.1f}% of 8 TB/s)")
            grad = torch.randn(16, B, 2, device=dev).to(dtype)
            ge = torch.zeros(total, 2, device=dev, dtype=dtype)
            for path in (1, 2):
                if path == 1 and B > (1 << 18):
                    continue
                G.set_backward_path(path)
                t = timeit(lambda: G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0), iters=5)
                print(f"grid_bwd[{('', 'atomics', 'binned')[path]}] {dtype} B={B}: {t*1e6:9.1f} us  {B/t/1e9:7.3f} Gpts/s  {B*bytes_pt/t/1e9:8.1f} GB/s algorithmic")
            G.set_backward_path(0)
    # coherent points (samples along rays) as in training
    grid, bits = syn.lego_like_density_grid(seed=0)
    bits = torch.from_numpy(bits).to(dev)
    for N in (4096, 16384, 640000):
        poses = syn.orbit_poses(1, seed=0)
        r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=N if N < 640000 else -1, generator=torch.Generator().manual_seed(0))
        ro, rd = r["rays_o"][0].contiguous().to(dev), r["rays_d"][0].contiguous().to(dev)
        N = ro.shape[0]
        aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
        nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
        t = timeit(lambda: R.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars))
        print(f"near_far N={N}: {t*1e6:.1f} us")
        M = N * 160
        xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, device=dev)

        def march():
            counter.zero_()
            R.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
        for path in (1, 2):
            if path == 2 and N > 16384:
                continue
            R.set_march_path(path)
            t = timeit(march)
            m = int(counter[0])
            print(f"march_rays_train[{'lane' if path == 1 else 'wave'}] N={N}: {t*1e6:.1f} us  samples={m} ({m/N:.1f}/ray)  {m/t/1e9:.3f} Gsamples/s")
        R.set_march_path(0)
        sig = torch.rand(m, device=dev) * 20
        rgb = torch.rand(m, 3, device=dev)
        ws, dp, im = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
        dl = deltas[:m].contiguous()
        t = timeit(lambda: R.composite_rays_train_forward(sig, rgb, dl, rays, m, N, 1e-4, ws, dp, im))
        print(f"composite_fwd N={N} M={m}: {t*1e6:.1f} us  {m*24/t/1e9:.1f} GB/s algorithmic")
        gs, gc = torch.zeros(m, device=dev), torch.zeros(m, 3, device=dev)
        gw, gi = torch.ones(N, device=dev), torch.ones(N, 3, device=dev)
        t = timeit(lambda: R.composite_rays_train_backward(gw, gi, sig, rgb, dl, rays, ws, im, m, N, 1e-4, gs, gc))
        print(f"composite_bwd N={N} M={m}: {t*1e6:.1f} us  {m*40/t/1e9:.1f} GB/s algorithmic")
        # grid fwd on ray-coherent samples
        if m > 0:
            xs = ((xyzs[:m] + 1) / 2).contiguous()
            emb = (torch.rand(total, 2, device=dev) * 2e-4 - 1e-4).half()
            out = torch.empty(16, m, 2, device=dev, dtype=torch.half)
            t = timeit(lambda: G.grid_encode_forward(xs, emb, offs, out, m, 3, 2, 16, S, 16, None, 0, False, 0))
            print(f"grid_fwd f16 ray-coherent B={m}: {t*1e6:.1f} us  {m/t/1e9:.3f} Gpts/s  {m*588/t/1e9:.1f} GB/s algorithmic")
    # ffmlp
    for (inn, W, n) in ((32, 64, 2), (32, 64, 3)):
        for B in (1 << 18, 1 << 21):
            x = torch.randn(B, inn, device=dev).half()
            w = (torch.rand(W * (inn + W * (n - 1) + 16), device=dev) - 0.5).half()
            fb = torch.empty(n, B, W, device=dev, dtype=torch.half)
            out = torch.empty(B, 16, device=dev, dtype=torch.half)
            t = timeit(lambda: F.ffmlp_forward(x, w, B, inn, 16, W, n, 0, 6, fb, out))
            flops = 2 * B * (inn * W + (n - 1) * W * W + 16 * W)
            byts = B * (2 * inn + 2 * n * W + 32)
            print(f"ffmlp_fwd in={inn} W={W} n={n} B={B}: {t*1e6:.1f} us  {flops/t/1e12:.1f} TFLOP/s  {byts/t/1e9:.0f} GB/s")
            t = timeit(lambda: F.ffmlp_inference(x, w, B, inn, 16, W, n, 0, 6, fb, out))
            print(f"ffmlp_inf in={inn} W={W} n={n} B={B}: {t*1e6:.1f} us  {flops/t/1e12:.1f} TFLOP/s")
            grad = torch.randn(B, 16, device=dev).half()
            bb = torch.zeros(n, B, W, device=dev, dtype=torch.half)
            gw = torch.zeros_like(w)
            gi = torch.zeros(B, inn, device=dev, dtype=torch.half)
            t = timeit(lambda: F.ffmlp_backward(grad, x, w, fb, B, inn, 16, W, n, 0, 6, True, bb, gi, gw))
            print(f"ffmlp_bwd in={inn} W={W} n={n} B={B}: {t*1e6:.1f} us  {3*flops/t/1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
