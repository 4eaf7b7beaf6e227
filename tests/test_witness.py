"""CPU: the C oracle (fp32, statement-by-statement restatement of the reference's kernels) against an INDEPENDENT float64
numpy statement of the same mathematics (oracle/witness.py) — a second witness for the two oracle files that no reference
fixture pins (raymarching, gridencoder).  Integers exact; floats within fp32 rounding of the float64 values."""
import numpy as np
import pytest
import torch

from oracle import witness as W


def _enc_meta(D, L, C, base, log2T, desired, align=False):
    pls = np.exp2(np.log2(desired / base) / (L - 1)) if L > 1 else 1.0
    offs, off = [], 0
    for i in range(L):
        res = int(np.ceil(base * pls ** i))
        n = min(2 ** log2T, (res if align else res + 1) ** D)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return np.array(offs, dtype=np.int32), float(np.log2(pls)), off


@pytest.mark.parametrize("D,L,C,base,log2T,desired,gridtype,align,smooth", [
    (3, 16, 2, 16, 19, 2048, "hash", False, False),   # Lego configuration
    (3, 6, 4, 8, 12, 128, "hash", False, True),
    (2, 5, 2, 8, 10, 128, "tiled", True, False),
    (3, 4, 1, 4, 9, 24, "tiled", False, False),
])
def test_grid_encode_oracle_vs_float64_witness(oracle, D, L, C, base, log2T, desired, gridtype, align, smooth):
    offsets, S, total = _enc_meta(D, L, C, base, log2T, desired, align)
    g = torch.Generator().manual_seed(L * 100 + D)
    B = 3000
    x = torch.rand(B, D, generator=g)
    x[:16] = torch.tensor([0.0, 1.0, 0.5][:D] + [0.25] * max(0, D - 3))  # faces of the unit cube
    x[16:24] = 1.0
    x[24:32] = -0.01  # outside
    table = torch.rand(total, C, generator=g) * 2 - 1
    out = torch.empty(L, B, C)
    oracle.GridBackend.grid_encode_forward(x, table, torch.from_numpy(offsets), out, B, D, C, L, S, base, None,
                                           0 if gridtype == "hash" else 1, align, 1 if smooth else 0)
    want, rows, weights = W.grid_encode(x.numpy(), table.numpy(), offsets, S, base, gridtype, align, smooth)
    got = out.permute(1, 0, 2).reshape(B, L * C).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=4e-6)  # |features| <= 1: a few fp32 ulps of an 8-term sum
    # rows: the oracle's index function on the witness' cells
    scale, res = W.level_geometry(L, S, base)
    for l in (0, L // 2, L - 1):
        size = int(offsets[l + 1] - offsets[l])
        cells = np.stack([np.random.default_rng(l).integers(0, int(res[l]) + 1, 64) for _ in range(D)], 1).astype(np.uint32)
        w_rows, _ = W.grid_rows(cells, size, int(res[l]), gridtype, align)
        o_rows = [oracle.grid_index(D, C, 0 if gridtype == "hash" else 1, align, 0, size, int(res[l]), c) // C for c in cells]
        assert w_rows.tolist() == [int(v) for v in o_rows]
    # backward: table gradient
    grad = torch.randn(L, B, C, generator=g)
    ge = torch.zeros(total, C)
    oracle.GridBackend.grid_encode_backward(grad, x, table, torch.from_numpy(offsets), ge, B, D, C, L, S, base, None, None,
                                            0 if gridtype == "hash" else 1, align, 1 if smooth else 0)
    want_g = W.grid_encode_backward(grad.permute(1, 0, 2).reshape(B, L * C).numpy(), rows, weights, offsets, total, C)
    np.testing.assert_allclose(ge.numpy(), want_g, rtol=1e-4, atol=1e-4 * np.abs(want_g).max())


def test_compositing_oracle_vs_float64_witness(oracle):
    g = torch.Generator().manual_seed(3)
    N = 200
    counts = torch.randint(0, 60, (N,), generator=g)
    counts[:5] = 0
    offs = torch.cumsum(counts, 0) - counts
    M = int(counts.sum())
    rays = torch.stack([torch.randperm(N, generator=g), offs, counts], 1).int().contiguous()
    sig = torch.rand(M, generator=g) * 60
    sig[: M // 10] = 0
    sig[M // 2: M // 2 + 50] = 5000.0  # opaque: early termination
    rgb = torch.rand(M, 3, generator=g)
    dt = torch.rand(M, generator=g) * 0.01 + 0.002
    deltas = torch.stack([dt, dt * (1 + 0.1 * torch.rand(M, generator=g))], 1).contiguous()
    ws, dp, im = torch.empty(N), torch.empty(N), torch.empty(N, 3)
    oracle.RaymarchingBackend.composite_rays_train_forward(sig, rgb, deltas, rays, M, N, 1e-4, ws, dp, im)
    w_ws, w_dp, w_im, _, kept = W.composite_train(sig.numpy(), rgb.numpy(), deltas.numpy(), rays.numpy(), 1e-4)
    assert (kept < counts.numpy()).any(), "some rays must terminate early"
    np.testing.assert_allclose(ws.numpy(), w_ws, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dp.numpy(), w_dp, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(im.numpy(), w_im, rtol=1e-4, atol=1e-6)
    g_ws, g_im = torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    gs, gc = torch.zeros(M), torch.zeros(M, 3)
    oracle.RaymarchingBackend.composite_rays_train_backward(g_ws, g_im, sig, rgb, deltas, rays, ws, im, M, N, 1e-4, gs, gc)
    w_gs, w_gc = W.composite_train_grads(g_ws.numpy().astype(np.float64), g_im.numpy(), sig.numpy(), rgb.numpy(), deltas.numpy(),
                                         rays.numpy(), 1e-4)
    np.testing.assert_allclose(gc.numpy(), w_gc, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gs.numpy(), w_gs, rtol=2e-4, atol=2e-4 * np.abs(w_gs).max())


def test_morton_and_packbits_oracle_vs_witness(oracle):
    g = torch.Generator().manual_seed(1)
    xyz = torch.randint(0, 128, (5000, 3), generator=g, dtype=torch.int32)
    idx = torch.empty(5000, dtype=torch.int32)
    oracle.RaymarchingBackend.morton3D(xyz, 5000, idx)
    assert np.array_equal(idx.numpy().astype(np.int64), W.morton3d(xyz.numpy()))
    grid = torch.rand(8192, generator=g)
    bits = torch.empty(1024, dtype=torch.uint8)
    oracle.RaymarchingBackend.packbits(grid, 1024, 0.37, bits)
    assert np.array_equal(bits.numpy(), W.packbits(grid.numpy(), np.float32(0.37)))
