#!/usr/bin/env python3
"""Do two grid backward ops overlap when they are issued on two streams?  (The scatter is bound by vector-instruction issue,
the accumulate by LDS atomics: profiles/r09_grid_backward.md.)  Times K ops back to back on one stream against K/2 + K/2 on
two streams, Lego configuration, ray-ordered points."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_grid as bg  # noqa: E402
import s3d_hip  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
    dev = "cuda"
    torch.manual_seed(0)
    G = s3d_hip.GridBackend
    offs, S, total = bg.grid_meta(dev)
    emb = (torch.rand(total, 2, device=dev) * 2 - 1).half()
    x = bg.ray_ordered_points(B, dev)
    grad = (torch.randn(16, B, 2, device=dev) * 1e-3).half()
    ge = [torch.zeros(total, 2, device=dev, dtype=torch.half) for _ in range(2)]
    s = [torch.cuda.Stream(), torch.cuda.Stream()]

    def op(i):
        G.grid_encode_backward(grad, x, emb, offs, ge[i], B, 3, 2, 16, S, 16, None, None, 0, False, 0)

    def run(two, K=20):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s[1].wait_stream(s[0])
        with torch.cuda.stream(s[0]):
            a.record()
        s[1].wait_stream(s[0])
        for k in range(K):
            i = k % 2
            with torch.cuda.stream(s[i if two else 0]):
                op(i)
        s[0].wait_stream(s[1])
        with torch.cuda.stream(s[0]):
            b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / K * 1e3

    for two in (False, True, False, True):
        run(two, 4)
        print(f"B={B} streams={'2' if two else '1'}: {run(two):.1f} us per backward op")


if __name__ == "__main__":
    main()
