"""The training marcher against the REFERENCE TEXT, bit for bit (tests/golden/march_kernels.npz).

`oracle/gen_golden.py march` transliterates `kernel_march_rays_train` (raymarching.cu:311-478: both passes, span reservation,
voxel skip) statement by statement and runs it ray by ray in thread order, with C's typing explicit and every float product
that feeds an add evaluated as one fused multiply-add (nvcc's default -fmad=true).  The oracle's `march_rays_train` — and the
HIP marcher, tests/test_gpu_golden.py — must reproduce the ray table, the counter and every sample exactly: the build's
restatement IS the reference's text under that contraction model."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

CASES = ("c1", "c2", "c1_noperturb")


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "march_kernels.npz"))


def run_march(R, G, tag, dev="cpu"):
    from nerf import synthetic as syn
    C, H, max_steps, M, N = G[f"{tag}_cfg"].tolist()
    bits = G[f"{tag}_bits"] if C > 1 else syn.lego_like_density_grid(seed=0)[1]
    t = lambda k: torch.from_numpy(np.ascontiguousarray(G[f"{tag}_{k}"])).to(dev)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rays = torch.zeros(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    R.march_rays_train(t("rays_o"), t("rays_d"), torch.from_numpy(np.ascontiguousarray(bits)).to(dev), float(G[f"{tag}_bound"]),
                       float(G[f"{tag}_dt_gamma"]), max_steps, N, C, H, M, t("nears"), t("fars"), xyzs, dirs, deltas, rays, counter,
                       t("noises"))
    return [v.cpu().numpy() for v in (rays, counter, xyzs, dirs, deltas)]


def check_march(got, G, tag):
    rays, counter, xyzs, dirs, deltas = got
    assert np.array_equal(counter, G[f"{tag}_counter"]), (counter, G[f"{tag}_counter"])
    assert np.array_equal(rays, G[f"{tag}_rays"]), np.argwhere(rays != G[f"{tag}_rays"])[:5]
    for name, a in (("xyzs", xyzs), ("dirs", dirs), ("deltas", deltas)):
        want = G[f"{tag}_{name}"]
        bad = np.argwhere(a.view(np.uint32) != want.view(np.uint32))
        assert bad.size == 0, (name, bad[:5], a[tuple(bad[0])], want[tuple(bad[0])])
    assert int(counter[0]) > 500 and (G[f"{tag}_rays"][:, 2] > 0).sum() >= 10


@pytest.mark.parametrize("tag", CASES)
def test_march_rays_train_reproduces_the_reference_text(oracle, G, tag):
    check_march(run_march(oracle.RaymarchingBackend, G, tag), G, tag)


def run_march_infer(R, G, dev="cpu"):
    from nerf import synthetic as syn
    C, H, max_steps = G["mi_cfg"].tolist()
    bits = torch.from_numpy(np.ascontiguousarray(syn.lego_like_density_grid(seed=0)[1])).to(dev)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(G[k])).to(dev)
    ro, rd, nears, fars = t("mi_rays_o"), t("mi_rays_d"), t("mi_nears"), t("mi_fars")
    out = []
    for it in range(2):
        alive, rays_t, noises = t(f"mi{it}_alive"), t(f"mi{it}_rays_t"), t(f"mi{it}_noises")
        n_alive, n_step = alive.shape[0], int(G[f"mi{it}_n_step"])
        rows = n_alive * n_step
        xyzs, dirs, deltas = torch.zeros(rows, 3, device=dev), torch.zeros(rows, 3, device=dev), torch.zeros(rows, 2, device=dev)
        R.march_rays(n_alive, n_step, alive, rays_t, ro, rd, 1.0, 0.0, max_steps, C, H, bits, nears, fars, xyzs, dirs, deltas, noises)
        out.append([v.cpu().numpy() for v in (xyzs, dirs, deltas)])
    return out


def check_march_infer(out, G):
    for it, (xyzs, dirs, deltas) in enumerate(out):
        for name, a in (("xyzs", xyzs), ("dirs", dirs), ("deltas", deltas)):
            want = G[f"mi{it}_{name}"]
            bad = np.argwhere(a.view(np.uint32) != want.view(np.uint32))
            assert bad.size == 0, (it, name, bad[:5], a[tuple(bad[0])], want[tuple(bad[0])])
        assert (deltas[:, 0] > 0).sum() > 500 and (deltas[:, 0] == 0).sum() > 30     # filled slots and unfilled ones (zeros)


def test_march_rays_inference_reproduces_the_reference_text(oracle, G):
    """kernel_march_rays (raymarching.cu:701-800), two loop iterations (8 and 12 slots per ray, the second perturbed): every
    slot — position, direction, both deltas, the zeros of the slots a ray does not fill — bit for bit"""
    check_march_infer(run_march_infer(oracle.RaymarchingBackend, G), G)
