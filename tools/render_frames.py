#!/usr/bin/env python3
"""Render-only workload for rocprofv3: `--save F` trains the bench model (400 graph-replayed steps on the synthetic scene) and
writes its state; `--load F --frames K` renders K 800x800 frames through NeRFRenderer.run_cuda's sync-free inference loop
(infer_batch_scale 4, sync_every 4: the bench's `render_mrays_per_s` configuration) and prints ms per frame."""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "seal-3d_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
from nerf import network_ff, synthetic as syn  # noqa: E402
from nerf.trainer import GraphedTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--load")
    ap.add_argument("--frames", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.save:
        from test_gpu_trainer import _setup, _run
        model, batches = _setup(n_rays=4096, n_batches=16)
        tr = GraphedTrainer(model, 4096, lr=1e-2, fp16=True)
        _run(tr, batches, 400)
        torch.cuda.synchronize()
        torch.save(model.state_dict(), a.save)
        print("saved", a.save)
        return
    model = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    model.load_state_dict(torch.load(a.load))
    model.eval()
    model.infer_batch_scale, model.sync_every = 4, 4
    poses = syn.orbit_poses(1, seed=0).cuda()
    r = syn.get_rays(poses[:1], syn.lego_intrinsics(), 800, 800)
    ro, rd = r["rays_o"].contiguous(), r["rays_d"].contiguous()

    def frame():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model.render(ro, rd, bg_color=1, perturb=False, max_steps=1024, dt_gamma=0, T_thresh=1e-4)
    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.frames):
        frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.frames
    print(f"render 800x800: {dt*1e3:.2f} ms/frame, {0.64/dt:.1f} Mrays/s over {a.frames} frames (+1 warm-up frame in the trace)")


if __name__ == "__main__":
    main()
