#!/usr/bin/env python3
"""Training-outcome PSNR over many seeds (bench.long_run_quality with --seeds N), one JSON per build:

    S3D_HIP_LIB=<variant.so> python tools/psnr_seeds.py --seeds 16 --tag pairs4 --out gpurun_out/psnr_pairs4.json

Both arrangements (native fp16 graph-replayed step | torch.optim.Adam on fp32 gradients, eager) train configs[1]'s network
from the same initial weights on the same batches for 3,000 steps per seed; PSNR on four held-out views.  Reported per
arrangement: mean, sample standard deviation, standard error; for the paired difference the same plus a 95 % interval
(Student t)."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "seal-3d_amd")]

T975 = {2: 12.706, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365, 9: 2.306, 10: 2.262, 12: 2.201, 16: 2.131,
        24: 2.069, 32: 2.040}


def t975(n):
    ks = sorted(T975)
    return T975[max([k for k in ks if k <= n] or [2])] if n < 60 else 1.96


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--hw", type=int, default=200)
    ap.add_argument("--tag", default="default")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    import s3d_hip
    from nerf import synthetic as syn
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    s3d_hip.lib()
    R = s3d_hip.RaymarchingBackend
    _, bits = syn.lego_like_density_grid(seed=0)
    scene_bits = torch.from_numpy(bits).to(dev)
    boxes = syn.lego_like_boxes(0)
    out = bench.long_run_quality(args, dev, R, scene_bits, boxes, steps=a.steps, seeds=a.seeds, n_views=a.views, hw=a.hw,
                                 note=lambda m: print("[psnr_seeds]", m, file=sys.stderr, flush=True))
    n = a.seeds
    for tag in ("native_fp16_graph", "torch_adam_fp32_eager"):
        out[tag]["psnr_db_sem"] = out[tag]["psnr_db_std"] / math.sqrt(n)
        out[tag].pop("runs", None)
    out["delta_db_sem"] = out["delta_db_std"] / math.sqrt(n)
    out["delta_db_ci95"] = [out["delta_db"] - t975(n) * out["delta_db_sem"], out["delta_db"] + t975(n) * out["delta_db_sem"]]
    out["build"] = {"tag": a.tag, "lib": os.path.relpath(s3d_hip.LIB_PATH, REPO)}
    line = json.dumps(out)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(line + "\n")


if __name__ == "__main__":
    main()
