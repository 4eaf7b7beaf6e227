"""FFMLP forward / backward timings (sigma and colour nets of network_ff) for the two backward paths."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
from tools.microbench import timeit
F = s3d_hip.FFMLPBackend
for (inn, W, n) in ((32, 64, 2), (32, 64, 3)):
    for B in [int(b) for b in os.environ.get('S3D_BENCH_SIZES', '').split(',') if b] or (1 << 17, 1 << 18, 1 << 20):
        x = torch.randn(B, inn, device="cuda").half()
        w = (torch.rand(W * (inn + W * (n - 1) + 16), device="cuda") - 0.5).half()
        fb = torch.empty(n, B, W, device="cuda", dtype=torch.half)
        out = torch.empty(B, 16, device="cuda", dtype=torch.half)
        t = timeit(lambda: F.ffmlp_forward(x, w, B, inn, 16, W, n, 0, 6, fb, out))
        t2 = timeit(lambda: F.ffmlp_forward(x, w, B, inn, 16, W, n, 0, 6, None, out))
        grad = torch.randn(B, 16, device="cuda").half()
        bb = torch.empty(n, B, W, device="cuda", dtype=torch.half)
        gw = torch.zeros_like(w)
        gi = torch.empty(B, inn, device="cuda", dtype=torch.half)
        t3 = timeit(lambda: F.ffmlp_backward(grad, x, w, fb, B, inn, 16, W, n, 0, 6, True, bb, gi, gw))
        t4 = timeit(lambda: F.ffmlp_backward(grad, x, w, None, B, inn, 16, W, n, 0, 6, True, None, gi, gw))
        t5 = timeit(lambda: F.ffmlp_backward(grad, x, w, None, B, inn, 16, W, n, 0, 6, False, None, None, gw))
        print(f"in={inn} W={W} n={n} B={B:8d}: fwd(train) {t*1e6:7.1f} us  fwd(no buffer) {t2*1e6:7.1f} us  bwd(2-kernel) {t3*1e6:7.1f} us  "
              f"bwd(fused) {t4*1e6:7.1f} us  bwd(fused, no dX) {t5*1e6:7.1f} us", flush=True)
