#!/usr/bin/env python3
"""How many of a training step's marched samples carry an exactly-zero gradient out of composite_rays_train_backward
(the samples behind a ray's early termination, raymarching.cu:560-573)?  Trains the bench model, then runs the step body
eagerly with the backward call wrapped.  Also prints the per-ray live-prefix statistics (live samples are a prefix of
every ray's span)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "seal-3d_amd")):
    sys.path.insert(0, p)
import s3d_hip  # noqa: E402
import bench  # noqa: E402
from nerf import network_ff, synthetic as syn  # noqa: E402
from nerf.trainer import GraphedTrainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
    tr = GraphedTrainer(model, 4096, lr=1e-2, fp16=True, update_extra_interval=16)
    R = s3d_hip.RaymarchingBackend
    _, bits = syn.lego_like_density_grid(seed=0)
    scene_bits = torch.from_numpy(bits).to(dev)
    batches, _ = bench.make_batches(32, 4096, 0, dev, R, scene_bits, syn.lego_like_boxes(0))
    for i in range(steps):
        tr.train_step(*batches[i % 32])
    torch.cuda.synchronize()
    model.mean_count = tr.budget
    model.train()
    seen = []
    orig = R.composite_rays_train_backward

    def wrapped(gws, gim, sigmas, rgbs, deltas, rays, ws, im, M, N, T, gs, gc):
        orig(gws, gim, sigmas, rgbs, deltas, rays, ws, im, M, N, T, gs, gc)
        r = rays.cpu().long()
        tot = int((r[:, 2]).sum())
        dead = ((gs[:M] == 0) & (gc[:M] == 0).all(-1))
        # live prefix per ray
        off, cnt = r[:, 1], r[:, 2]
        d = dead.cpu()
        live = 0
        prefix_ok = True
        for o, c in zip(off.tolist()[:512], cnt.tolist()[:512]):
            seg = d[o:o + c]
            nl = int((~seg).sum())
            live += nl
            if nl and bool(seg[:nl].any()):
                prefix_ok = False
        seen.append((M, tot, int(dead[:tot].sum()) if tot <= M else -1, prefix_ok))

    R.composite_rays_train_backward = staticmethod(wrapped)
    for _ in range(3):
        tr._body_fb()
        tr._body_opt()
    torch.cuda.synchronize()
    for M, tot, dead, ok in seen:
        print(f"M={M} samples={tot} dead={dead} ({100.0 * dead / max(tot, 1):.1f} %) live-is-a-prefix(first 512 rays)={ok}")


if __name__ == "__main__":
    main()
