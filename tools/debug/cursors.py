import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd")); sys.path.insert(0, os.path.join(REPO, "tools"))
import s3d_hip
from bench_grid import grid_meta, ray_ordered_points
dev = "cuda"
G = s3d_hip.GridBackend
offs, S, total = grid_meta(dev)
B = 1 << 18
x = ray_ordered_points(B, dev)
emb = torch.zeros(total, 2, device=dev, dtype=torch.half)
grad = (torch.randn(16, B, 2, device=dev) * 1e-3).half()
grad[:, torch.rand(B, device=dev) < 0.2] = 0
ge = torch.zeros(total, 2, device=dev, dtype=torch.half)
G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0)
torch.cuda.synchronize()
ws = list(s3d_hip._ws.buf.values())[0]
cur = ws[256:256 + 16 * 64 * 8 * 4].view(torch.int32).view(16, 64, 8).cpu().numpy()
ovn = ws[256 + 16 * 64 * 8 * 4: 256 + 16 * 64 * 8 * 4 + 16 * 64 * 4].view(torch.int32).view(16, 64).cpu().numpy()
np.set_printoptions(linewidth=200)
for l in (0, 1, 2, 4, 8, 12, 15):
    print("level", l, "records per sub-bucket (sum over slices):", cur[l].sum(0), " per slice min/max:", cur[l].sum(1).min(), cur[l].sum(1).max(), " spills", ovn[l].sum())
print("total records", cur.sum(), "per point", cur.sum() / B)
