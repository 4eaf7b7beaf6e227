import os, sys, argparse, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip, bench
from nerf import network_ff, synthetic as syn
from nerf.trainer import Trainer, GraphedTrainer, psnr
dev = torch.device("cuda")
_, bits = syn.lego_like_density_grid(seed=0)
bits = torch.from_numpy(bits).to(dev)
boxes = syn.lego_like_boxes(0)
R = s3d_hip.RaymarchingBackend
pool, _ = bench.make_batches(768, 4096, 4242, dev, R, bits, boxes)
views = syn.orbit_poses(4, seed=977)
rays = [syn.get_rays(views[i:i + 1].to(dev), syn.lego_intrinsics(200, 200), 200, 200) for i in range(4)]
gts = [bench.analytic_targets(r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous(), bits, boxes, R) for r in rays]
kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
torch.manual_seed(5)
init = network_ff.NeRFNetwork(**kw).to(dev).state_dict()
which = sys.argv[1:] or ["native", "torch"]
steps = 3000
for tag in which:
    torch.manual_seed(6)
    m = network_ff.NeRFNetwork(**kw).to(dev)
    m.load_state_dict(init)
    native = tag == "native"
    tr = GraphedTrainer(m, 4096, lr=1e-2, fp16=True) if native else Trainer(m, lr=1e-2, fp16=True, native_optim=False)
    for i in range(steps):
        lr = 1e-2 * 0.1 ** min(i / steps, 1.0)
        for g in tr.optimizer.param_groups:
            g["lr"] = lr
        if native and tr.graph is not None and i % 100 == 0:
            tr.graph = None
        loss = tr.train_step(*pool[i % len(pool)])
        if i % 250 == 0 or i == steps - 1:
            vals = [psnr(tr.render_image(r["rays_o"].contiguous(), r["rays_d"].contiguous())["image"][0], gt) for r, gt in zip(rays[:1], gts[:1])]
            print(tag, i, "loss", float(loss), "scale", tr.scaler.get_scale(), "mean_count", m.mean_count, "psnr0", round(vals[0], 2), flush=True)
    del tr, m
