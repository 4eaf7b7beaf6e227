import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
from tools.microbench import grid_meta
G = s3d_hip.GridBackend
offs, S, total = grid_meta()
B = 1 << 17
for dtype in (torch.float16,):
    emb = (torch.rand(total, 2, device="cuda") * 2e-4 - 1e-4).to(dtype)
    x = torch.rand(B, 3, device="cuda")
    grad = torch.randn(16, B, 2, device="cuda").to(dtype)
    ge = torch.zeros(total, 2, device="cuda", dtype=dtype)
    out = torch.empty(16, B, 2, device="cuda", dtype=dtype)
    for path in (2,):
        G.set_backward_path(path)
        for _ in range(3):
            G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0)
torch.cuda.synchronize()
