import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import s3d_hip as hip
from test_gpu_gridencoder import _enc_meta, _inputs
D, L, C, base, log2T, desired, gridtype, align, interp, dtype = (3, 8, 1, 16, 15, 512, 1, True, 0, torch.float32)
offsets, S, total = _enc_meta(D, L, C, base, log2T, desired, align)
B = 4096 + 37
x = _inputs(B, D, seed=D * 100 + L).cuda()
g = torch.Generator().manual_seed(1)
emb = ((torch.rand(total, C, generator=g) * 2 - 1)).to(dtype).cuda()
torch.rand(1, generator=g)
grad = torch.randn(L, B, C, generator=g).to(dtype).cuda()
og = offsets.cuda()
res = {}
for path in (1, 3, 2):
    hip.GridBackend.set_backward_path(path)
    ge = torch.zeros(total, C, dtype=dtype, device="cuda")
    hip.GridBackend.grid_encode_backward(grad, x, emb, og, ge, B, D, C, L, S, base, None, None, gridtype, align, interp)
    res[path] = ge.cpu()
cidx = torch.empty(B, L, 2 ** D, dtype=torch.int32, device="cuda")
hip.GridBackend.grid_corner_indices(x, og, cidx, B, D, C, L, S, base, gridtype, align)
cidx = cidx.cpu()
d = (res[2] - res[3]).abs().squeeze(-1)
bad = torch.nonzero(d > 1e-4).squeeze(-1)
print("gen2 vs atomics max", float((res[3] - res[1]).abs().max()), " gen3 vs gen2 max", float(d.max()), " nbad", bad.numel())
offs = offsets.tolist()
for r in bad.tolist()[:40]:
    lvl = max(l for l in range(L) if offs[l] <= r)
    loc = r - offs[lvl]
    pts = torch.nonzero((cidx[:, lvl, :] == loc).any(-1)).squeeze(-1).tolist()
    print(f"row {r} level {lvl} local {loc}: gen3 {float(res[2][r,0]):+.5f} gen2 {float(res[3][r,0]):+.5f} diff {float(res[2][r,0]-res[3][r,0]):+.5f} points {pts[:12]} lanes {[p % 64 for p in pts[:12]]}")
    for p in pts[:6]:
        k = torch.nonzero(cidx[p, lvl] == loc).squeeze(-1).tolist()
        print("      pt", p, "x", x[p].tolist(), "corner", k, "grad", float(grad[lvl, p, 0]))
print("---- explain")
scales = hip.level_scales(L, S, base)
xc = x.cpu()
for lvl in range(2):
    sc = scales[lvl]
    pos = xc * sc + (0.0 if align else 0.5)
    pg = pos.floor().int()
    fr = pos - pg
    same = (pg[1:] == pg[:-1]).all(-1)
    idx = torch.nonzero(same).squeeze(-1) + 1
    print("level", lvl, "adjacent same-cell pairs (second index):", idx.tolist()[:40], "lanes", [(i % 64) for i in idx.tolist()[:40]])
for r in bad.tolist():
    lvl = max(l for l in range(L) if offs[l] <= r)
    loc = r - offs[lvl]
    sc = scales[lvl]
    pos = xc * sc + (0.0 if align else 0.5)
    pg = pos.floor().int(); fr = pos - pg
    dd = float(res[2][r, 0] - res[3][r, 0])
    pts = torch.nonzero((cidx[:, lvl, :] == loc).any(-1)).squeeze(-1).tolist()
    for p in pts:
        for k in torch.nonzero(cidx[p, lvl] == loc).squeeze(-1).tolist():
            w = 1.0
            for d_ in range(D):
                w *= float(fr[p, d_]) if (k >> d_) & 1 else 1 - float(fr[p, d_])
            c = w * float(grad[lvl, p, 0])
            if abs(c + dd) < 1e-4 * max(1, abs(dd)) or abs(c - dd) < 1e-4 * max(1, abs(dd)):
                print(f"row {r} lvl {lvl}: diff {dd:+.5f} == {'-' if abs(c+dd)<abs(c-dd) else '+'} contribution of pt {p} (lane {p%64}, thread {p%512}) corner {k}")
