import os, sys, argparse
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip, bench
from nerf import network_ff, synthetic as syn
from nerf.trainer import Trainer, psnr
dev = torch.device("cuda")
_, bits = syn.lego_like_density_grid(seed=0)
bits = torch.from_numpy(bits).to(dev)
boxes = syn.lego_like_boxes(0)
R = s3d_hip.RaymarchingBackend
args = argparse.Namespace(num_rays=4096, seed=0)
pool, _ = bench.make_batches(64, 4096, 4242, dev, R, bits, boxes)
kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
order = sys.argv[1] if len(sys.argv) > 1 else "torch"
for tag in ([ "native", "torch"] if order == "both" else [order]):
    torch.manual_seed(6)
    m = network_ff.NeRFNetwork(**kw).to(dev)
    tr = Trainer(m, lr=1e-2, fp16=True, native_optim=(tag == "native"))
    for i in range(400):
        loss = tr.train_step(*pool[i % len(pool)])
        if i % 50 == 0:
            sc = tr.scaler.get_scale()
            print(tag, i, "loss", float(loss), "scale", sc, "mean_count", m.mean_count, flush=True)
