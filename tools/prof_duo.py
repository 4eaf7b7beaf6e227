"""Shader-clock stamps of one pair (compute wave + weight-gradient wave) of k_ffmlp_backward_duo.
Library built with -DS3D_FFMLP_PROF=<NH> (1: density network, 2: colour network):  S3D_HIP_LIB=<variant> python tools/prof_duo.py <NH>"""
import ctypes as C, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
from tools.microbench import timeit
F = s3d_hip.FFMLPBackend
NH = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 269824
inn, W, n = 32, 64, NH + 1
x = torch.randn(B, inn, device="cuda").half()
w = (torch.rand(W * (inn + W * (n - 1) + 16), device="cuda") - 0.5).half()
grad = torch.randn(B, 16, device="cuda").half()
gw = torch.zeros_like(w)
gi = torch.empty(B, inn, device="cuda", dtype=torch.half)
run = lambda: F.ffmlp_backward(grad, x, w, None, B, inn, 16, W, n, 0, 6, True, None, gi, gw)
t = timeit(run)
print(f"NH={NH} B={B}: backward (fused) {t*1e6:.1f} us")
lib = s3d_hip.lib()
buf = np.zeros((2, 1024), dtype=np.uint64)
lib.s3d_debug_ffmlp_prof_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 1)
run()
lib.s3d_debug_ffmlp_prof_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 0)
per = {0: 5 + 2 * (NH + 1), 1: 2 + 2 * (NH + 1)}
names = {0: ["load+L0", "recompute", "stage0", "bar"] + sum([[f"st{j+1}", "bar"] for j in range(NH + 1)], []),
         1: ["drain", "bar"] + sum([[f"use{j}", "bar"] for j in range(NH + 1)], [])}
for role, rn in ((0, "compute"), (1, "weight-gradient")):
    t = buf[role].astype(np.int64)
    k = int((t > 0).sum())
    p = per[role]
    rounds = k // p
    if rounds < 2:
        print(rn, "no stamps", k); continue
    a = t[:rounds * p].reshape(rounds, p)
    d = np.diff(np.concatenate([a, np.roll(a[:, :1], -1, axis=0)], axis=1), axis=1)[:-1]  # last column: to the next round's first stamp
    print(f"== {rn} wave: {rounds} rounds, {int(a[-1, -1] - a[0, 0])} ticks from first to last stamp; mean ticks per segment (rounds 1..):")
    m = d[1:].mean(0)
    for nm, v in zip(names[role] + ["loop"], m):
        print(f"   {nm:10s} {v:8.0f}")
    print(f"   per round  {m.sum():8.0f}")
