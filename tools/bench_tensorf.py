"""TensoRF VM features (BASELINE config 5 shapes: rank 16x3 / 48x3, resolution 128 and 300): fused forward kernel vs the
reference's grid_sample sequence, and the backward (grid_sample sequence in both cases)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
from tensoRF import network as trf  # noqa: E402
from tools.microbench import timeit  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
    only_res = [int(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else (128, 300)
    modes = [m == "fused" for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else (True, False)
    for res in only_res:
        torch.manual_seed(0)
        net = trf.NeRFNetwork(resolution=[res] * 3, bound=1, cuda_ray=True).cuda()
        x = (torch.rand(N, 3, device="cuda") * 2 - 1)
        for fused in modes:
            net.fused_vm = fused

            def fwd():
                with torch.no_grad():
                    return net.get_sigma_feat(x), net.get_color_feat(x)

            def fwd_bwd():
                net.zero_grad(set_to_none=True)
                s, c = net.get_sigma_feat(x), net.get_color_feat(x)
                (s.sum() + c.sum()).backward()
            tf, tb = timeit(fwd, iters=10), timeit(fwd_bwd, iters=5)
            print(f"res {res:3d} N={N} {'fused' if fused else 'torch'}: forward {tf*1e6:8.1f} us   forward+backward {tb*1e6:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
