"""The remaining native kernels of the extension packages against the REFERENCE TEXT (tests/golden/encoder_kernels.npz).

`oracle/gen_golden.py enc` transliterates `kernel_freq` / `kernel_freq_backward` (freqencoder.cu:30-94), `kernel_sph_from_ray`
(raymarching.cu:163-198) and `kernel_grad_tv` (gridencoder.cu:503-607, fp32 tables) statement by statement and runs them thread by
thread with numpy float32 scalars (C typing explicit, nvcc's multiply-add contraction modelled).  `__sinf` is CUDA's approximate
intrinsic (2^-21.4 absolute error on [-pi, pi]): float32 sin stands in for it and the comparison is at 2e-6 absolute; the
total-variation gradient is a normalised sum whose atomics have no defined order: 2e-6 of the largest entry, and the SET of rows
it touches exactly.  The HIP kernels: tests/test_gpu_golden.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "encoder_kernels.npz"))


def run_freq(F, G, tag, dev="cpu"):
    x, g = torch.from_numpy(G[f"freq_{tag}_x"]).to(dev), torch.from_numpy(G[f"freq_{tag}_grad"]).to(dev)
    B, D = x.shape
    deg = int(G[f"freq_{tag}_deg"])
    C = D + 2 * D * deg
    y = torch.full((B, C), 9.0, device=dev)
    F.freq_encode_forward(x.contiguous(), B, D, deg, C, y)
    gx = torch.full((B, D), 9.0, device=dev)
    # (the backward reads the reference's stored outputs, as autograd hands the forward's own over)
    F.freq_encode_backward(g.contiguous(), torch.from_numpy(G[f"freq_{tag}_y"]).to(dev), B, D, deg, C, gx)
    return y.cpu().numpy(), gx.cpu().numpy()


def check_freq(got, G, tag):
    y, gx = got
    want_y, want_g = G[f"freq_{tag}_y"], G[f"freq_{tag}_grad_x"]
    D = G[f"freq_{tag}_x"].shape[1]
    assert np.array_equal(y[:, :D].view(np.int32), want_y[:, :D].view(np.int32)), "identity columns: bit for bit (incl. -0)"
    np.testing.assert_allclose(y, want_y, rtol=0, atol=2e-6)
    np.testing.assert_allclose(gx, want_g, rtol=2e-6, atol=2e-6 * float(np.abs(want_g).max()))


def run_sph(R, G, dev="cpu"):
    ro, rd = torch.from_numpy(G["sph_rays_o"]).to(dev), torch.from_numpy(G["sph_rays_d"]).to(dev)
    N = ro.shape[0]
    coords = torch.full((N, 2), 9.0, device=dev)
    R.sph_from_ray(ro.contiguous(), rd.contiguous(), float(G["sph_radius"]), N, coords)
    return coords.cpu().numpy()


def check_sph(got, G):
    np.testing.assert_allclose(got, G["sph_coords"], rtol=0, atol=2e-6)


def run_tv(Gr, G, tag, dev="cpu"):
    D, C, gridtype, ac, L, H = [int(v) for v in G[f"tv_{tag}_cfg"]]
    x, emb = torch.from_numpy(G[f"tv_{tag}_x"]).to(dev), torch.from_numpy(G[f"tv_{tag}_emb"]).to(dev)
    offs = torch.from_numpy(G[f"tv_{tag}_offsets"]).to(dev)
    g = torch.zeros_like(emb)
    Gr.grad_total_variation(x.contiguous(), emb.contiguous(), g, offs, float(G[f"tv_{tag}_weight"]), x.shape[0], D, C, L,
                            float(G[f"tv_{tag}_S"]), H, gridtype, bool(ac))
    return g.cpu().numpy()


def check_tv(got, G, tag):
    want = G[f"tv_{tag}_grad"]
    assert np.array_equal(np.abs(got).sum(1) > 0, np.abs(want).sum(1) > 0), "rows touched"
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * float(np.abs(want).max()))


@pytest.mark.parametrize("tag", ["tensorf", "dirs"])
def test_oracle_freq_encoder_vs_reference_text(oracle, G, tag):
    check_freq(run_freq(oracle.FreqBackend, G, tag), G, tag)


def test_oracle_sph_from_ray_vs_reference_text(oracle, G):
    check_sph(run_sph(oracle.RaymarchingBackend, G), G)


@pytest.mark.parametrize("tag", ["hash", "tiled_ac"])
def test_oracle_grad_total_variation_vs_reference_text(oracle, G, tag):
    check_tv(run_tv(oracle.GridBackend, G, tag), G, tag)
