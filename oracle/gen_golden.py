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
import zlib

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


# ----------------------------------------------------------------------------- reference wrappers on the oracle
def _install_reference_stack():
    """Make the reference's Python importable on CPU: its five `_backend` extension modules are replaced by the
    CPU oracle, GPU-only third-party imports by empty stubs, and `.cuda()` by the identity."""
    from oracle import oracle_backend as ob
    ob.build()

    def mod(name, obj=None, **attrs):
        m = types.ModuleType(name)
        if obj is not None:
            for k in dir(obj):
                if not k.startswith("_"):
                    setattr(m, k, getattr(obj, k))
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("_raymarching", ob.RaymarchingBackend)
    mod("_gridencoder", ob.GridBackend)
    mod("_shencoder", ob.SHBackend)
    mod("_freqencoder", ob.FreqBackend)
    mod("_ffmlp", ob.FFMLPBackend)
    for name in ("trimesh", "cv2", "tensorboardX", "lpips", "mcubes", "imageio", "torch_ema", "rich", "rich.console",
                 "packaging", "scipy.spatial.transform", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod(name)
    mod("turtle", backward=None, forward=None)  # stray import in the reference's ffmlp.py:2 (needs tkinter)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    if REF not in sys.path:
        sys.path.insert(0, REF)


def _seeded(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def gen_wrappers():
    _install_reference_stack()
    import importlib
    out = {}
    # ---- gridencoder/grid.py
    grid = importlib.import_module("gridencoder.grid")
    for tag, kw in (("hash", dict(input_dim=3, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=8, per_level_scale=2)),
                    ("smooth", dict(input_dim=2, num_levels=3, level_dim=4, base_resolution=8, log2_hashmap_size=10,
                                    desired_resolution=64, interpolation="smoothstep")),
                    ("tiled_ac", dict(input_dim=3, num_levels=3, level_dim=1, base_resolution=8, log2_hashmap_size=9,
                                      desired_resolution=32, gridtype="tiled", align_corners=True))):
        enc = grid.GridEncoder(**kw)
        emb = _seeded(enc.embeddings.shape, 11, -1, 1)
        enc.embeddings.data.copy_(emb)
        x = _seeded((257, kw["input_dim"]), 12, -1.05, 1.05).requires_grad_(True)
        y = enc(x, bound=1)
        go = _seeded(y.shape, 13, -1, 1)
        y.backward(go)
        out.update({f"grid_{tag}_offsets": enc.offsets.numpy(), f"grid_{tag}_emb": emb.numpy(), f"grid_{tag}_x": x.detach().numpy(),
                    f"grid_{tag}_y": y.detach().numpy(), f"grid_{tag}_go": go.numpy(),
                    f"grid_{tag}_gemb": enc.embeddings.grad.numpy(), f"grid_{tag}_gx": x.grad.numpy(),
                    f"grid_{tag}_pls": np.float64(enc.per_level_scale)})
    lego = grid.GridEncoder(desired_resolution=2048)
    out["grid_lego_offsets"] = lego.offsets.numpy()
    # ---- shencoder / freqencoder / ffmlp modules
    sh = importlib.import_module("shencoder.sphere_harmonics")
    d = torch.nn.functional.normalize(_seeded((300, 3), 21, -1, 1), dim=-1).requires_grad_(True)
    ysh = sh.SHEncoder(degree=4)(d)
    ysh.backward(_seeded(ysh.shape, 22, -1, 1))
    out.update(sh_d=d.detach().numpy(), sh_y=ysh.detach().numpy(), sh_gd=d.grad.numpy())
    fq = importlib.import_module("freqencoder.freq")
    xf = _seeded((100, 3), 23, -1, 1).requires_grad_(True)
    yf = fq.FreqEncoder(input_dim=3, degree=4)(xf)
    yf.backward(_seeded(yf.shape, 24, -1, 1))
    out.update(freq_x=xf.detach().numpy(), freq_y=yf.detach().numpy(), freq_gx=xf.grad.numpy())
    ff = importlib.import_module("ffmlp.ffmlp")
    net = ff.FFMLP(32, 3, 64, 3)
    xin = (_seeded((200, 32), 25, -1, 1) * 0.5).half()
    net.train()
    with torch.autocast("cpu", enabled=False):
        w16 = net.weights.detach().half()
        # the reference wrapper only casts under CUDA autocast; call the Function with explicit halfs
        yy = ff.ffmlp_forward(torch.cat([xin, torch.zeros(56, 32, dtype=torch.half)]), w16.clone().requires_grad_(True), 32, 16, 64, 3,
                              0, 6, False, False)
    out.update(ffmlp_w=net.weights.detach().numpy(), ffmlp_x=xin.numpy(), ffmlp_y=yy.detach()[:200, :3].float().numpy(),
               ffmlp_num_parameters=np.int64(net.num_parameters), ffmlp_padded_out=np.int64(net.padded_output_dim))
    # ---- raymarching wrappers + renderer control flow
    rm = importlib.import_module("raymarching.raymarching")
    sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
    from nerf import synthetic as syn
    dens, bits = syn.lego_like_density_grid(seed=0)
    bits_t = torch.from_numpy(bits)
    poses = syn.orbit_poses(1, seed=0)
    r = syn.get_rays(poses, syn.lego_intrinsics(64, 64), 64, 64)
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    nears, fars = rm.near_far_from_aabb(ro, rd, torch.tensor([-1.0, -1, -1, 1, 1, 1]), 0.2)
    counter = torch.zeros(2, dtype=torch.int32)
    torch.manual_seed(5)
    xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, 1.0, bits_t, 1, 128, nears, fars, counter, -1, True, 128, False, 0, 1024)
    out.update(march_ro=ro.numpy(), march_rd=rd.numpy(), march_nears=nears.numpy(), march_fars=fars.numpy(),
               march_counter=counter.numpy(), march_rays=rays.numpy(), march_xyzs_shape=np.array(xyzs.shape),
               march_xyzs_sum=xyzs.double().sum(0).numpy(), march_deltas_sum=deltas.double().sum(0).numpy(),
               march_xyzs_head=xyzs[:256].numpy(), march_deltas_head=deltas[:256].numpy())
    # renderer with a deterministic analytic "network"
    renderer = importlib.import_module("nerf.renderer")
    lo, hi = syn.lego_like_boxes(0)

    class Analytic(renderer.NeRFRenderer):
        def forward(self, x, dd):
            sig = syn.box_density(x, lo, hi, sigma=40.0)
            rgb = (x * 0.5 + 0.5).clamp(0, 1) * (0.5 + 0.5 * dd.abs())
            return sig, rgb

        def density(self, x):
            return {"sigma": syn.box_density(x, lo, hi, sigma=40.0)}
    R = Analytic(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    R.train()
    torch.manual_seed(7)
    R.update_extra_state()      # full sweep
    tr = R.run_cuda(ro[None], rd[None], bg_color=1, perturb=True, max_steps=1024)
    torch.manual_seed(8)
    R.update_extra_state()
    R.eval()
    ev = R.run_cuda(ro[None], rd[None], bg_color=1, perturb=False, max_steps=1024)
    out.update(rend_bitfield=R.density_bitfield.numpy().copy(), rend_mean_density=np.float64(R.mean_density),
               rend_mean_count=np.int64(R.mean_count), rend_iter_density=np.int64(R.iter_density),
               rend_train_image=tr["image"][0].numpy(), rend_train_depth=tr["depth"][0].numpy(),
               rend_eval_image=ev["image"][0].numpy(), rend_eval_depth=ev["depth"][0].numpy(),
               rend_step_counter=R.step_counter.numpy().copy())
    # reference network (nerf/network.py): parameter names/shapes and a forward on fixed weights
    network = importlib.import_module("nerf.network")
    network.NeRFNetwork._self = network.NeRFNetwork
    torch.manual_seed(3)
    net = network.NeRFNetwork(bound=1, cuda_ray=True, log2_hashmap_size=14)
    for k, p in net.named_parameters():
        p.data.copy_(_seeded(p.shape, zlib.crc32(k.encode()) % 1000, -0.5, 0.5))
    xq = _seeded((64, 3), 31, -1, 1)
    dq = torch.nn.functional.normalize(_seeded((64, 3), 32, -1, 1), dim=-1)
    sg, cl = net(xq, dq)
    out.update(net_param_names=np.array([k for k, _ in net.named_parameters()]),
               net_param_shapes=np.array([str(tuple(p.shape)) for _, p in net.named_parameters()]),
               net_seeds=np.array([zlib.crc32(k.encode()) % 1000 for k, _ in net.named_parameters()]),
               net_x=xq.numpy(), net_d=dq.numpy(), net_sigma=sg.detach().numpy(), net_color=cl.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **out)
    print("wrappers: wrote wrappers.npz with", len(out), "arrays")


SECTIONS = {"sh": gen_sh, "wrappers": gen_wrappers}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or list(SECTIONS)
    for w in which:
        SECTIONS[w]()
