#!/usr/bin/env python3
"""Hash-grid forward: what ONE level of the Lego configuration costs the XCD that serves it (GPU box) — the input of the
level -> XCD plan (csrc/gridencoder.hip: balance_forward_plan).  Needs a calibration build of the library
(`tools/build_variants.sh gridencoder "planenv:-DS3D_FWD_PLAN_ENV"`, run with S3D_HIP_LIB=<that .so>): with
S3D_FWD_ONLY_LEVEL=l the kernel serves level l alone, all of it on its home XCD l % 8, for all B points — the unit the plan moves."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "seal-3d_amd")); sys.path.insert(0, os.path.join(%r, "tools"))
import torch, s3d_hip
from bench_grid import grid_meta, ray_ordered_points, timeit
dev = "cuda"
G = s3d_hip.GridBackend
offs, S, total = grid_meta(dev)
emb = (torch.rand(total, 2, device=dev) * 2 - 1).half()
B = 1 << 18
out = torch.empty(16, B, 2, device=dev, dtype=torch.half)
res = []
for order, x in (("ray", ray_ordered_points(B, dev)), ("random", torch.rand(B, 3, device=dev))):
    res.append(timeit(lambda: G.grid_encode_forward(x, emb, offs, out, B, 3, 2, 16, S, 16, None, 0, False, 0), 40) * 1e6)
print("RESULT %%.1f %%.1f" %% tuple(res))
''' % (REPO, REPO, REPO)


def main():
    print("level   ray us  random us   (B = 2^18, the level alone on its home XCD)")
    for l in list(range(16)) + [None]:
        env = dict(os.environ)
        if l is None:
            env.pop("S3D_FWD_ONLY_LEVEL", None)
        else:
            env["S3D_FWD_ONLY_LEVEL"] = str(l)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300).stdout
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT")]
        print(f"{'all' if l is None else l:>5} " + (line[0][7:] if line else "failed"), flush=True)


if __name__ == "__main__":
    main()
