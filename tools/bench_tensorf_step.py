"""BASELINE config 5 (TensoRF VM-48, resolution 128 and 300) training step on the synthetic Lego-shaped scene: the build's
fused VM kernels vs the reference's grid_sample sequence (same trainer, same marching / compositing kernels)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
from nerf import synthetic as syn  # noqa: E402
from tensoRF import network as trf  # noqa: E402
from tensoRF.utils import GraphedTrainer, Trainer  # noqa: E402


def main():
    poses = syn.orbit_poses(8, seed=0).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    only_res = [int(r) for r in sys.argv[1].split(",")] if len(sys.argv) > 1 else (128, 300)
    modes = [m == "fused" for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else (True, False)
    # trainer kinds (tensoRF/utils.py; every step carries the L1 penalty of the reference's step, weight 1e-4):
    #   torch = eager, torch.optim.Adam + GradScaler; native = eager, NativeAdam; graph = HIP-graph replay, NativeAdam
    kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["torch"]
    for res in only_res:
        for fused, kind in [(f, k) for f in modes for k in kinds]:
            torch.manual_seed(0)
            net = trf.NeRFNetwork(resolution=[res] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
            net.fused_vm = fused
            net.density_grid.copy_(torch.from_numpy(grid))
            net.density_bitfield.copy_(torch.from_numpy(bits))
            net.iter_density = 100
            if kind == "graph":
                tr = GraphedTrainer(net, 4096, lr0=2e-2, lr1=1e-3, fp16=True, update_extra_interval=10 ** 9)
            else:
                tr = Trainer(net, lr0=2e-2, lr1=1e-3, fp16=True, update_extra_interval=10 ** 9, native_optim=(kind == "native"))
            tr.global_step = 1
            batches = []
            for k in range(8):
                r = syn.get_rays(poses[k:k + 1], syn.lego_intrinsics(), 800, 800, N=4096, generator=torch.Generator().manual_seed(k))
                batches.append((r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous(), torch.rand(4096, 3, device="cuda")))
            for k in range(4):  # warm-up
                tr.train_step(*batches[k % 8])
            # sample budget M as the reference has it after its first grid update: mean of the step counters
            net.mean_count = int(net.step_counter[:4, 0].float().mean().item())
            net.local_step = 0
            for k in range(4):
                tr.train_step(*batches[k % 8])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 16
            for k in range(steps):
                tr.train_step(*batches[k % 8])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            print(f"tensoRF VM-48 res {res}: {'fused VM kernels' if fused else 'grid_sample sequence'}, {kind} trainer: {dt*1e3:7.2f} ms/step, "
                  f"{net.mean_count} samples/step", flush=True)


if __name__ == "__main__":
    main()
