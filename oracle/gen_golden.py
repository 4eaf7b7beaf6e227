#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference (AUTHORING CONTAINER ONLY).

TEST INFRASTRUCTURE.  This script is the only place that reads /root/reference.
It never copies reference source into the repo: it *executes* reference code /
evaluates reference expressions on seeded inputs and stores inputs + expected
outputs as small fixtures.  /root/reference does not exist on the GPU box, so
nothing under tests/ imports this file; the fixtures it wrote are committed.

Sections
  sh        shencoder/src/shencoder.cu:49-355 — every `outputs[i] = <expr>` /
            `dx|dy|dz[i] = <expr>` line is evaluated with numpy float32 scalars
            semantics (one rounding per operator, no contraction) on 1,024
            seeded unit vectors + 64 non-unit vectors → PINS the SH oracle and
            the HIP kernel for degree 1..8 incl. the Jacobian.
  sh_torch  testing/test_shencoder.py:8-89 `SHEncoder_torch` (degree <= 5),
            executed as is.
  mlp       testing/test_ffmlp.py:11-43 bias-free torch `MLP` twin with the
            seed-42 init (ffmlp/ffmlp.py:141-144) → pins ffmlp dense math.
  freq      encoding.py:5-43 torch `FreqEncoder` (sin/cos of 2^f x) → pins the
            freq encoder values (the CUDA kernel's column order is re-derived
            from freqencoder.cu:45-57 and checked in the test).
  wrappers  the reference's own Python wrappers (gridencoder/grid.py,
            raymarching/raymarching.py, shencoder/sphere_harmonics.py,
            freqencoder/freq.py, ffmlp/ffmlp.py, nerf/renderer.py run_cuda /
            update_extra_state, nerf/network.py) imported with their
            `_backend` replaced by the CPU oracle: pins the HOST logic (offset
            tables, padding, counters, zero-init contracts, control flow) that
            the build's wrappers must reproduce.  The native arithmetic under
            them is the oracle's — for raymarching/gridencoder parity with the
            CUDA build stays UNPINNED (no reference fixtures exist).
"""
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _f32_expr(expr):
    """C float expression -> python expression on float32 numpy arrays."""
    expr = re.sub(r"(\d+\.\d+(?:e[+-]?\d+)?)f", r"F(\1)", expr)
    return expr


def gen_sh():
    src = open(os.path.join(REF, "shencoder/src/shencoder.cu")).read().split("\n")
    rows = {"outputs": {}, "dx": {}, "dy": {}, "dz": {}}
    pat = re.compile(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*(.*?)\s*;")
    for line in src[48:355]:
        m = pat.match(line)
        if m:
            rows[m.group(1)][int(m.group(2))] = m.group(3)
    assert all(len(v) == 64 for v in rows.values()), {k: len(v) for k, v in rows.items()}
    g = torch.Generator().manual_seed(1234)
    d = torch.randn(1024, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    extra = torch.rand(64, 3, generator=g) * 2 - 1  # non-unit, inside [-1,1]^3
    axes = torch.tensor([[1., 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [0, 0, 0]])
    pts = torch.cat([d, extra, axes]).numpy().astype(np.float32)
    F = np.float32
    x, y, z = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
    env = dict(F=F, x=x, y=y, z=z)
    # shencoder.cu:303 uses pow(z, 3): float pow(float,int), correctly rounded here
    env["pow"] = lambda a, b: (a.astype(np.float64) ** b).astype(np.float32)
    # shencoder.cu:44-47 temporaries, float32 products
    env.update(xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z)
    env["xyz"] = env["xy"] * z
    env.update(x4=env["x2"] * env["x2"], y4=env["y2"] * env["y2"], z4=env["z2"] * env["z2"])
    env.update(x6=env["x4"] * env["x2"], y6=env["y4"] * env["y2"], z6=env["z4"] * env["z2"])
    res = {}
    for name, tab in rows.items():
        arr = np.zeros((pts.shape[0], 64), np.float32)
        for i, e in tab.items():
            v = eval(_f32_expr(e), {"__builtins__": {}}, env)
            arr[:, i] = np.asarray(v, dtype=np.float32)
        res[name] = arr
    np.savez_compressed(os.path.join(OUT, "sh_deg8.npz"), inputs=pts, outputs=res["outputs"],
                        dx=res["dx"], dy=res["dy"], dz=res["dz"])
    print("sh: wrote sh_deg8.npz", pts.shape)

    # SHEncoder_torch (degree <= 5), executed as is
    lines = open(os.path.join(REF, "testing/test_shencoder.py")).read().split("\n")
    ns = {}
    exec("import torch\nimport torch.nn as nn\n" + "\n".join(lines[7:89]), ns)
    dd = torch.from_numpy(pts[:1024])
    out5 = ns["SHEncoder_torch"](degree=5)(dd).numpy()
    np.savez_compressed(os.path.join(OUT, "sh_torch_deg5.npz"), inputs=pts[:1024], outputs=out5)
    print("sh_torch: wrote sh_torch_deg5.npz")


SECTIONS = {"sh": gen_sh}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or list(SECTIONS)
    for w in which:
        SECTIONS[w]()
