#!/usr/bin/env python3
"""Which torch ops does one training step launch?  Trains the bench model to steady state, then runs the graphed trainer's
step body eagerly under torch.profiler and prints every device kernel with the aten op (and input shapes) that launched it."""
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
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
    tr = GraphedTrainer(model, 4096, lr=1e-2, fp16=True, update_extra_interval=16)
    R = s3d_hip.RaymarchingBackend
    _, bits = syn.lego_like_density_grid(seed=0)
    scene_bits = torch.from_numpy(bits).to(dev)
    batches, _ = bench.make_batches(32, 4096, 0, dev, R, scene_bits, syn.lego_like_boxes(0))
    for i in range(200):
        tr.train_step(*batches[i % 32])
    torch.cuda.synchronize()
    saved = model.mean_count
    model.mean_count = tr.budget
    model.train()
    for _ in range(2):
        tr._body_fb(); tr._body_opt()
    torch.cuda.synchronize()
    ues = "--ues" in sys.argv
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        if ues:  # the occupancy sweep instead of the step
            with torch.autocast("cuda", dtype=torch.float16):
                model.partial_grid_update_device()
        else:
            tr._body_fb()
            tr._body_opt()
        torch.cuda.synchronize()
    model.mean_count = saved
    # kernel -> launching op via the correlation the profiler keeps: walk CPU ops, list their device kernels
    if ues:
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
        return
    rows = []
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.kernels:
            for k in ev.kernels:
                rows.append((k.duration, k.name[:70], ev.name, str(ev.input_shapes)[:90]))
    seen = set()
    print(f"{'us':>7}  kernel | op | shapes")
    for d, k, op, sh in rows:
        key = (k, op, sh, round(d, 1))
        if key in seen:
            continue
        seen.add(key)
        print(f"{d:7.1f}  {k} | {op} | {sh}")


if __name__ == "__main__":
    main()
