#!/usr/bin/env python3
"""Time the one-launch network pair (s3d_ffmlp_ngp_pair_inference) alone: NGP and Seal variant, training-step and
render-iteration batch sizes.  `S3D_PAIR_CAP` (workgroups) and variant libraries (`S3D_HIP_LIB`) for A/B runs."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "seal-3d_amd")):
    sys.path.insert(0, p)
import s3d_hip  # noqa: E402


def main():
    F = s3d_hip.FFMLPBackend
    torch.manual_seed(0)
    dev = "cuda"
    out = []
    for B in (280064, 1600000 // 128 * 128):
        for seal in (False, True):
            e0 = (torch.randn(16, B, 2, device=dev) * 0.3).half()
            e1 = (torch.randn(16, B, 2, device=dev) * 0.3).half() if seal else None
            nls, nlc = (2, 2) if seal else (2, 3)
            ws = (torch.randn(64 * 32 + 64 * 64 * (nls - 1) + 16 * 64, device=dev) * 0.1).half()
            wc = (torch.randn(64 * (64 if seal else 32) + 64 * 64 * (nlc - 1) + 16 * 64, device=dev) * 0.1).half()
            d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
            sigma, rgb = torch.empty(B, device=dev), torch.empty(B, 3, device=dev)

            def run():
                F.ngp_pair_inference(e0, ws, wc, B, 64, nls, nlc, d, sigma, rgb, 1, None, None, None, e1)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            b.record()
            torch.cuda.synchronize()
            out.append(f"{'seal' if seal else 'ngp '} B={B}: {a.elapsed_time(b) / 20 * 1e3:7.1f} us")
    print(f"cap={os.environ.get('S3D_PAIR_CAP', 'default')} lib={os.path.basename(os.environ.get('S3D_HIP_LIB', 'shipped'))}: " + " | ".join(out))


if __name__ == "__main__":
    main()
