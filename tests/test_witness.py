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


def _rays_and_scene(n, seed, cascade, bound):
    """rays aimed at a box scene from outside (the construction of tests/test_gpu_raymarching.py, kept local)"""
    from nerf import synthetic as syn
    g = torch.Generator().manual_seed(seed)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    o = -d * (2.2 * bound) + (torch.rand(n, 3, generator=g) - 0.5) * 0.8 * bound
    d[0] = torch.tensor([1.0, 0.0, 0.0])      # axis-parallel: two of the reciprocal directions are infinite
    o[0] = torch.tensor([-3.0 * bound, 0.1, -0.2])
    d[1] = torch.tensor([0.0, 0.0, -1.0])
    o[1] = torch.tensor([0.3, 0.2, 3.0 * bound])
    o[2] = torch.tensor([5.0 * bound, 5.0 * bound, 0.0])  # misses the box
    d[2] = torch.tensor([0.0, 0.0, 1.0])
    H = 128
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    base, _ = syn.lego_like_density_grid(seed=seed)
    grid[0] = base.reshape(-1)[: H ** 3] if base.size >= H ** 3 else base.reshape(-1)
    rng = np.random.default_rng(seed)
    for c in range(1, cascade):
        grid[c] = (rng.random(H ** 3) < 0.02) * 20.0
    bits = np.packbits((grid.reshape(-1) > 10.0).astype(np.uint8), bitorder="little")
    return o.contiguous(), d.contiguous(), torch.from_numpy(bits)


@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("cascade,bound,dt_gamma,max_steps", [(1, 1.0, 0.0, 1024), (2, 2.0, 1.0 / 128, 1024), (3, 4.0, 0.0, 4096),
                                                              (1, 1.0, 0.0, 20)])
def test_marcher_oracle_vs_independent_witness(oracle, perturb, cascade, bound, dt_gamma, max_steps):
    """near_far_from_aabb and the two-pass DDA of march_rays_train: the C oracle against the numpy statement written from the
    CUDA text alone (oracle/witness.py) — per-ray counts, span offsets, counters, sample positions, directions and deltas,
    all bit for bit, on the four marcher configurations of the GPU parity test."""
    N, M = 160, 160 * 160
    ro, rd, bits = _rays_and_scene(N, seed=cascade, cascade=cascade, bound=bound)
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
    nears, fars = torch.empty(N), torch.empty(N)
    oracle.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
    w_near, w_far = W.near_far_from_aabb(ro.numpy(), rd.numpy(), aabb.numpy(), 0.2)
    assert np.array_equal(nears.numpy().view(np.uint32), w_near.view(np.uint32))
    assert np.array_equal(fars.numpy().view(np.uint32), w_far.view(np.uint32))
    assert float(nears[2]) == np.finfo(np.float32).max and float(nears[0]) < 10
    g = torch.Generator().manual_seed(5)
    noises = torch.rand(N, generator=g) if perturb else torch.zeros(N)
    xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
    rays = torch.empty(N, 3, dtype=torch.int32)
    counter = torch.zeros(2, dtype=torch.int32)
    oracle.RaymarchingBackend.march_rays_train(ro, rd, bits, bound, dt_gamma, max_steps, N, cascade, 128, M, nears, fars, xyzs, dirs,
                                               deltas, rays, counter, noises)
    wx, wd, wdl, wrays, wcount = W.march_rays_train(ro.numpy(), rd.numpy(), bits.numpy(), bound, dt_gamma, max_steps, cascade, 128, M,
                                                    nears.numpy(), fars.numpy(), noises.numpy())
    assert np.array_equal(rays.numpy(), wrays), "per-ray (id, offset, count) differ"
    assert np.array_equal(counter.numpy(), wcount)
    assert int(counter[0]) > N, "the scene must produce samples"
    if max_steps == 20:
        assert int(rays[:, 2].max()) == 20
    for got, want in ((xyzs, wx), (dirs, wd), (deltas, wdl)):
        assert np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("cascade,bound,dt_gamma,n_step", [(1, 1.0, 0.0, 5), (2, 2.0, 1.0 / 128, 8), (1, 1.0, 0.0, 1)])
def test_inference_iteration_oracle_vs_independent_witness(oracle, cascade, bound, dt_gamma, n_step):
    """march_rays + composite_rays (one iteration of the run_cuda eval loop, raymarching.cu:701-895): the C oracle against the
    numpy statement of the same iteration — sample rows bit for bit (incl. the zero rows of unfilled slots), the kill
    pattern identical, accumulated weights / depth / colour within fp32 rounding of the float64 values; three iterations, each
    on the survivors with their advanced t."""
    N = 200
    ro, rd, bits = _rays_and_scene(N, seed=cascade + 3, cascade=cascade, bound=bound)
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
    nears, fars = torch.empty(N), torch.empty(N)
    oracle.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
    g = torch.Generator().manual_seed(9)
    alive = torch.arange(N, dtype=torch.int32)
    rays_t = nears.clone()
    ws, dp, im = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    w_alive, w_t = alive.numpy().copy(), rays_t.numpy().astype(np.float64)
    w_ws, w_dp, w_im = np.zeros(N), np.zeros(N), np.zeros((N, 3))
    killed = unfilled = 0
    for it in range(3):
        n_alive = alive.shape[0]
        M = n_alive * n_step
        noises = torch.rand(n_alive, generator=g) if it == 0 else torch.zeros(n_alive)
        xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        oracle.RaymarchingBackend.march_rays(n_alive, n_step, alive, rays_t, ro, rd, bound, dt_gamma, 1024, cascade, 128, bits, nears,
                                             fars, xyzs, dirs, deltas, noises)
        wx, wd, wdl = W.march_rays(n_alive, n_step, alive.numpy(), rays_t.numpy(), ro.numpy(), rd.numpy(), bits.numpy(), bound,
                                   dt_gamma, 1024, cascade, 128, fars.numpy(), noises.numpy())
        for got, want in ((xyzs, wx), (dirs, wd), (deltas, wdl)):
            assert np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))
        unfilled += int((deltas[:, 0] == 0).sum())
        assert (deltas[:, 0] != 0).any()
        sig = torch.rand(M, generator=g) * 40
        rgb = torch.rand(M, 3, generator=g)
        # the witness works on ITS OWN state (float64 t), fed the same samples
        w_alive_it, w_t, w_ws, w_dp, w_im = W.composite_rays(n_alive, n_step, 1e-2, w_alive, w_t, sig.numpy(), rgb.numpy(),
                                                              deltas.numpy(), w_ws, w_dp, w_im)
        oracle.RaymarchingBackend.composite_rays(n_alive, n_step, 1e-2, alive, rays_t, sig, rgb, deltas, ws, dp, im)
        assert np.array_equal(alive.numpy(), w_alive_it), "kill pattern differs"
        np.testing.assert_allclose(ws.numpy(), w_ws, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(dp.numpy(), w_dp, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(im.numpy(), w_im, rtol=2e-5, atol=1e-6)
        keep = alive >= 0
        np.testing.assert_allclose(rays_t.numpy()[alive[keep].long().numpy()], w_t[w_alive_it[w_alive_it >= 0]], rtol=2e-6)
        killed += int((~keep).sum())
        alive = alive[keep].contiguous()
        w_alive = w_alive_it[w_alive_it >= 0]
        # (the marcher of the next iteration starts from the ORACLE's float32 t on both sides: the witness checks one
        #  iteration's arithmetic at a time)
        w_t = rays_t.numpy().astype(np.float64)
        if alive.numel() == 0:
            break
    assert killed > 0 and (unfilled > 0 or n_step == 1)
