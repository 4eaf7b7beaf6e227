"""The hash-grid forward against the REFERENCE TEXT (tests/golden/grid_kernels.npz).

`oracle/gen_golden.py grid` transliterates `kernel_grid` (gridencoder.cu:87-242) statement by statement — C typing explicit,
nvcc's multiply-add contraction modelled, `exp2f` correctly rounded — and runs it thread by thread for fp32 tables on the
reference's own encoder configurations (hash / smoothstep 2-D / tiled + align_corners / the Lego table).  The oracle's
`grid_encode_forward` — and the HIP forward, tests/test_gpu_golden.py — must reproduce the outputs bit for bit and the
Jacobian `dy_dx` to 1e-6 of its largest entry."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

TAGS = ("hash", "smooth", "tiled_ac", "lego")


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "grid_kernels.npz"))


def table(G, tag):
    """the fp32 table of a case: stored, or (Lego: 12.2 M rows) re-drawn from the generator's seeded stream"""
    D, C, gridtype, ac, interp, L, H = G[f"{tag}_cfg"].tolist()
    if G[f"{tag}_emb"].size:
        return G[f"{tag}_emb"]
    rng = np.random.default_rng(int(G[f"{tag}_emb_seed"]))
    x = rng.uniform(0, 1, G[f"{tag}_x"].shape)      # (the points were drawn first)
    return rng.uniform(-1, 1, (int(G[f"{tag}_offsets"][-1]), C)).astype(np.float32)


def run_forward(Gb, G, tag, dev="cpu"):
    D, C, gridtype, ac, interp, L, H = G[f"{tag}_cfg"].tolist()
    x = torch.from_numpy(G[f"{tag}_x"]).to(dev)
    emb = torch.from_numpy(table(G, tag)).to(dev)
    offsets = torch.from_numpy(G[f"{tag}_offsets"]).to(dev)
    B = x.shape[0]
    out = torch.full((L, B, C), 7.0, device=dev)
    jac = torch.full((B, L * D * C), 7.0, device=dev)
    Gb.grid_encode_forward(x, emb, offsets, out, B, D, C, L, float(G[f"{tag}_S"]), H, jac, gridtype, bool(ac), interp)
    return out.cpu().numpy(), jac.cpu().numpy()


def check_forward(got, G, tag):
    out, jac = got
    want, wjac = G[f"{tag}_outputs"], G[f"{tag}_dy_dx"]
    bad = np.argwhere(out.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, (tag, bad[:5], out[tuple(bad[0])], want[tuple(bad[0])])
    assert (want[:, 6:8] == 0).all() and np.abs(want).max() > 0.5          # the two out-of-range points: exact zeros
    np.testing.assert_allclose(jac, wjac, rtol=0, atol=1e-6 * float(np.abs(wjac).max()))


@pytest.mark.parametrize("tag", TAGS)
def test_grid_forward_reproduces_the_reference_text(oracle, G, tag):
    check_forward(run_forward(oracle.GridBackend, G, tag), G, tag)


def run_backward(Gb, G, tag, dev="cpu"):
    D, C, gridtype, ac, interp, L, H = G[f"{tag}_cfg"].tolist()
    x = torch.from_numpy(G[f"{tag}_x"]).to(dev)
    emb = torch.from_numpy(table(G, tag)).to(dev)
    offsets = torch.from_numpy(G[f"{tag}_offsets"]).to(dev)
    grad = torch.from_numpy(G[f"{tag}_grad"]).to(dev)
    jac = torch.from_numpy(G[f"{tag}_dy_dx"]).to(dev)
    B = x.shape[0]
    g_emb = torch.zeros_like(emb)
    g_in = torch.zeros(B, D, device=dev)
    Gb.grid_encode_backward(grad, x, emb, offsets, g_emb, B, D, C, L, float(G[f"{tag}_S"]), H, jac, g_in, gridtype, bool(ac), interp)
    return g_emb.cpu().numpy(), g_in.cpu().numpy()


def check_backward(got, G, tag):
    g_emb, g_in = got
    want, win = G[f"{tag}_grad_emb"], G[f"{tag}_grad_inputs"]
    assert np.array_equal(g_emb == 0, want == 0)          # exactly the rows some in-range point touches
    # (the reference adds with fp32 atomics in arrival order; the fixture applied them in thread order: rounding of the sums differs)
    np.testing.assert_allclose(g_emb, want, rtol=0, atol=2e-6 * float(np.abs(want).max()))
    np.testing.assert_allclose(g_in, win, rtol=0, atol=2e-6 * float(np.abs(win).max()))


@pytest.mark.parametrize("tag", ["hash", "smooth", "tiled_ac"])
def test_grid_backward_within_fp32_summation_order_of_the_reference_text(oracle, G, tag):
    """kernel_grid_backward (gridencoder.cu:245-337, fp32 branch) and kernel_input_backward (:340-366): the same set of
    `w * grad` contributions on the same rows, the input gradient from the forward's own dy_dx"""
    check_backward(run_backward(oracle.GridBackend, G, tag), G, tag)


def run_forward_f16(Gb, G, tag, dev="cpu"):
    D, C, gridtype, ac, interp, L, H = G[f"{tag}_cfg"].tolist()
    x = torch.from_numpy(G[f"{tag}_x"]).to(dev)
    emb = torch.from_numpy((table(G, tag) * np.float32(0.5)).astype(np.float16)).to(dev)   # (as the generator made the fp16 table)
    offsets = torch.from_numpy(G[f"{tag}_offsets"]).to(dev)
    B = x.shape[0]
    out = torch.full((L, B, C), 7.0, dtype=torch.float16, device=dev)
    jac = torch.full((B, L * D * C), 7.0, dtype=torch.float16, device=dev)
    Gb.grid_encode_forward(x, emb, offsets, out, B, D, C, L, float(G[f"{tag}_S"]), H, jac, gridtype, bool(ac), interp)
    return out.cpu().numpy(), jac.cpu().numpy()


def check_forward_f16(got, G, tag):
    out, jac = got
    want, wjac = G[f"{tag}_outputs_f16"], G[f"{tag}_dy_dx_f16"]
    bad = np.argwhere(out.view(np.uint16) != want.view(np.uint16))
    assert bad.size == 0, (tag, len(bad), bad[:5], out[tuple(bad[0])], want[tuple(bad[0])])
    assert np.abs(want.astype(np.float32)).max() > 0.2
    np.testing.assert_allclose(jac.astype(np.float32), wjac.astype(np.float32), rtol=0, atol=2e-3 * float(np.abs(wjac.astype(np.float32)).max()))


@pytest.mark.parametrize("tag", ["hash", "lego"])
def test_grid_forward_fp16_tables_reproduce_the_reference_text(oracle, G, tag):
    """the `-O` instantiation (scalar_t = at::Half): every `results[ch] += w * grid[..]` rounds the product to half and the sum
    to half (c10::Half has no `Half += float`: the right-hand side goes through Half's constructor) — outputs bit for bit"""
    check_forward_f16(run_forward_f16(oracle.GridBackend, G, tag), G, tag)


def run_backward_f16(Gb, G, tag, dev="cpu", **extra):
    D, C, gridtype, ac, interp, L, H = G[f"{tag}_cfg"].tolist()
    x = torch.from_numpy(G[f"{tag}_x"]).to(dev)
    emb = torch.from_numpy((table(G, tag) * np.float32(0.5)).astype(np.float16)).to(dev)
    offsets = torch.from_numpy(G[f"{tag}_offsets"]).to(dev)
    grad = torch.from_numpy(G[f"{tag}_grad_f16"]).to(dev)
    B = x.shape[0]
    g_emb = torch.zeros_like(emb)
    Gb.grid_encode_backward(grad, x, emb, offsets, g_emb, B, D, C, L, float(G[f"{tag}_S"]), H, None, None, gridtype, bool(ac), interp, **extra)
    return g_emb.cpu().numpy()


def check_backward_f16(got, G, tag, exact_sum):
    """`exact_sum`: the implementation adds the reference's per-contribution half values `(__half)(w * grad)` WITHOUT rounding in
    between (the HIP kernels' fixed-point sums): it must reproduce the correctly rounded exact sum — a conversion through fp32 on
    the way may double-round a near-tie, one ulp on a vanishing share of the entries.  Otherwise (an implementation that adds in
    fp16 like the reference's `__half2` atomics, in some order): within the drift thread order itself shows against the exact sum."""
    exact, rounded, thread = G[f"{tag}_grad_emb_f16_exact"], G[f"{tag}_grad_emb_f16_exact_rounded"], G[f"{tag}_grad_emb_f16_thread_order"]
    assert np.array_equal(got == 0, rounded == 0), "rows touched"
    if exact_sum:
        bad = got.view(np.uint16) != rounded.view(np.uint16)
        assert bad.mean() < 2e-3, int(bad.sum())
        ulp = np.abs(got.astype(np.float64) - exact) <= np.abs(np.spacing(rounded.astype(np.float32).astype(np.float16)).astype(np.float64))
        assert ulp.all()
    else:
        drift = np.abs(thread.astype(np.float64) - exact).max()
        assert np.abs(got.astype(np.float64) - exact).max() <= 2.0 * drift + 1e-12


def test_grid_backward_fp16_contributions_of_the_reference_text(oracle, G):
    """the `-O` branch of kernel_grid_backward (gridencoder.cu:321-327): per contribution `(__half)(w * grad)`, summed by `__half2`
    atomics in an order the hardware picks.  The fixture holds the exact sum of those half values and the outcome of thread order;
    the oracle (fp32 accumulation of the rounded contributions) stays within the drift thread order shows"""
    check_backward_f16(run_backward_f16(oracle.GridBackend, G, "hash"), G, "hash", exact_sum=False)
