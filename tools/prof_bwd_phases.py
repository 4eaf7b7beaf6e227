"""Per-phase shader-clock stamps of the third-generation binned backward (library built with -DS3D_BIN3_PROF)."""
import ctypes as C, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd")); sys.path.insert(0, os.path.join(REPO, "tools"))
import s3d_hip
from bench_grid import grid_meta, ray_ordered_points
dev = "cuda"
G = s3d_hip.GridBackend
offs, S, total = grid_meta(dev)
B = 1 << 18
order = sys.argv[1] if len(sys.argv) > 1 else "ray"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
x = torch.rand(B, 3, device=dev) if order == "random" else ray_ordered_points(B, dev)
emb = torch.zeros(total, 2, device=dev, dtype=torch.half)
grad = (torch.randn(16, B, 2, device=dev) * 1e-3).half()
if order == "ray":
    grad[:, (torch.arange(B, device=dev) % 64) >= 51] = 0
ge = torch.zeros(total, 2, device=dev, dtype=torch.half)
lib = s3d_hip.lib()
buf = np.zeros((2, 16384, 8), dtype=np.uint64)
def run():
    G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0)
for _ in range(3):
    run()
lib.s3d_debug_prof_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_size_t(buf.nbytes), 1)
run()
lib.s3d_debug_prof_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_size_t(buf.nbytes), 0)
chunks = (B + P - 1) // P
for kern, name, nph, nwg in ((0, "scatter", 8, chunks * 16), (1, "accumulate", 6, 1024)):
    t = buf[kern, :nwg, :nph].astype(np.int64)
    ok = t[:, 0] > 0
    t0 = t[ok, 0].min()
    span = (t[ok].max() - t0)
    print(f"== {name}: {ok.sum()} workgroups stamped, kernel span {span} ticks")
    lv = (np.arange(nwg) // (chunks if kern == 0 else 64))
    for l in range(16):
        m = ok & (lv == l) & (t[:, nph - 1] > 0)
        if not m.any():
            continue
        d = np.diff(t[m], axis=1)
        print(f"  level {l:2d}: n={m.sum():4d} start {int((t[m,0]-t0).mean()):8d}  phases " + " ".join(f"{int(v):7d}" for v in d.mean(0)) + f"  total {int((t[m,-1]-t[m,0]).mean()):7d}")
