import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
from nerf import network_ff, synthetic as syn
from nerf.trainer import GraphedTrainer
import bench
torch.manual_seed(0)
dev = torch.device("cuda")
model = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
tr = GraphedTrainer(model, 4096, lr=1e-2, fp16=True, update_extra_interval=16)
grid, bits = syn.lego_like_density_grid(seed=0)
batches, poses = bench.make_batches(8, 4096, 0, dev, s3d_hip.RaymarchingBackend, torch.from_numpy(bits).to(dev), syn.lego_like_boxes(0))
for i in range(200):
    tr.train_step(*batches[i % 8])
torch.cuda.synchronize()
tr.update_extra_interval = 10 ** 9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(64):
        tr.train_step(*batches[i % 8])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    g0 = time.perf_counter()
    for i in range(64):
        tr.graph.replay()
    torch.cuda.synchronize()
    g1 = time.perf_counter()
    print(f"fused={model.fused_head}: train_step {1e3*(t1-t0)/64:.3f} ms, bare graph replay {1e3*(g1-g0)/64:.3f} ms, budget {tr.budget}", flush=True)
