"""GPU: `n_valid` (seal3d_hip.h) — the ray marcher's device-side sample count handed to every per-sample kernel of the
training step.  Contract under test: for the rows that hold samples, and for every reduced gradient, a call with the
pointer equals a call without it on the same buffers with zeros in the tail; rows past round_up(*n_valid, 128) are
neither read (they are poisoned with NaN here) nor written (sentinels survive)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _counter(n):
    return torch.tensor([n, 0], dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("B,count", [(128 * 80, 5000), (128 * 80, 128 * 80 + 77), (128 * 40, 0), (128 * 80, 128 * 33)])
def test_network_ff_padded_batch_equals_zero_tail(hip, B, count):
    import s3d_hip
    from nerf import network_ff
    torch.manual_seed(0)
    net = network_ff.NeRFNetwork(bound=1, cuda_ray=True).cuda().train()
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    g = torch.Generator().manual_seed(1)
    live = min(count, B)
    n_eff = min(B, (count + 127) // 128 * 128)
    x = (torch.rand(B, 3, generator=g) * 2 - 1).cuda()
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).cuda()
    x[live:], d[live:] = 0, 0  # what march_rays_train leaves behind the samples (zero-filled buffers)
    w_s, w_c = torch.rand(B, generator=g).cuda(), torch.rand(B, 3, generator=g).cuda()
    w_s[live:], w_c[live:] = 0, 0  # composite_rays_train_backward: zero gradient for rows no ray owns

    def run(limit):
        net.zero_grad(set_to_none=True)
        xi, di = x.clone(), d.clone()
        if limit is not None:
            xi[n_eff:], di[n_eff:] = float("nan"), float("nan")  # absent rows must not be read
        with torch.autocast("cuda", dtype=torch.float16):
            if limit is not None:
                with s3d_hip.row_limit(limit, B):
                    sigma, rgb = net(xi, di)
            else:
                sigma, rgb = net(xi, di)
            loss = (sigma[:n_eff].float() * w_s[:n_eff]).sum() * 1e-3 + (rgb[:n_eff].float() * w_c[:n_eff]).sum()
        if n_eff:
            loss.backward()
        grads = [None if p.grad is None else p.grad.clone() for p in
                 (net.encoder.embeddings, net.sigma_net.weights, net.color_net.weights)]
        return sigma[:n_eff].detach().clone(), rgb[:n_eff].detach().clone(), grads

    ref = run(None)
    got = run(_counter(count))
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    for a, b, name in zip(ref[2], got[2], ("table", "sigma_net", "color_net")):
        if n_eff == 0:
            assert b is None or not b.any(), name
            continue
        assert torch.isfinite(b).all(), name
        assert torch.equal(a, b), f"{name}: max diff {(a - b).abs().max()}"
    # a limit announced for another batch size is ignored
    with torch.autocast("cuda", dtype=torch.float16), s3d_hip.row_limit(_counter(128), B + 128):
        sigma, _ = net(x, d)
    assert torch.equal(sigma[:n_eff], ref[0])


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("B", [4096, 128 * 100])  # direct-atomic and binned backward
def test_grid_kernels_leave_absent_rows_untouched(hip, dtype, B):
    import s3d_hip
    from tools.microbench import grid_meta
    G = s3d_hip.GridBackend
    offs, S, total = grid_meta()
    g = torch.Generator().manual_seed(2)
    count = B // 2 + 5
    n_eff = (count + 127) // 128 * 128
    x = torch.rand(B, 3, generator=g).cuda()
    emb = ((torch.rand(total, 2, generator=g) - 0.5)).cuda().to(dtype)
    grad = (torch.randn(16, B, 2, generator=g) * 1e-3).cuda().to(dtype)
    xp, gp = x.clone(), grad.clone()
    xp[n_eff:], gp[:, n_eff:] = float("nan"), float("nan")
    x0, g0 = x.clone(), grad.clone()
    x0[n_eff:], g0[:, n_eff:] = 0.5, 0
    nv = _counter(count)
    out_ref = torch.empty(16, B, 2, device="cuda", dtype=dtype)
    out = torch.full((16, B, 2), 7.0, device="cuda", dtype=dtype)
    G.grid_encode_forward(x0, emb, offs, out_ref, B, 3, 2, 16, S, 16, None, 0, False, 0)
    G.grid_encode_forward(xp, emb, offs, out, B, 3, 2, 16, S, 16, None, 0, False, 0, n_valid=nv)
    assert torch.equal(out[:, :n_eff], out_ref[:, :n_eff]) and bool((out[:, n_eff:] == 7.0).all())
    ge_ref, ge = torch.zeros_like(emb), torch.zeros_like(emb)
    G.grid_encode_backward(g0, x0, emb, offs, ge_ref, B, 3, 2, 16, S, 16, None, None, 0, False, 0)
    G.grid_encode_backward(gp, xp, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0, n_valid=nv)
    assert torch.isfinite(ge.float()).all()
    if B >= 8192:
        assert torch.equal(ge, ge_ref)  # binned path: deterministic
    else:
        torch.testing.assert_close(ge.float(), ge_ref.float(), rtol=1e-2, atol=1e-5)  # atomics: order varies


@pytest.mark.parametrize("fused", [True, False])
def test_ffmlp_padded_batch(hip, fused, monkeypatch):
    import ffmlp.ffmlp as ff
    monkeypatch.setattr(ff, "_FUSED_BACKWARD", fused)
    net = ff.FFMLP(32, 16, 64, 2).cuda().train()
    B, count = 128 * 64, 128 * 20 + 3
    n_eff = 128 * 21
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, 32, generator=g) * 0.5).cuda().half()
    w = torch.randn(B, 16, generator=g).cuda()
    x[n_eff:], w[n_eff:] = 0, 0
    res = []
    for nv in (None, _counter(count)):
        xi = x.clone().requires_grad_(True)
        if nv is not None:
            with torch.no_grad():
                xi[n_eff:] = float("nan")
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = net.forward_padded(xi, n_valid=nv)
        (y[:n_eff].float() * w[:n_eff]).sum().backward()
        res.append((y[:n_eff].detach().clone(), net.weights.grad.clone(), xi.grad[:n_eff].clone()))
    for a, b in zip(*res):
        assert torch.isfinite(b.float()).all() and torch.equal(a, b)


def test_inference_loop_skips_dead_slots_without_changing_the_image(hip, monkeypatch):
    """`live` of s3d_grid_encode_forward: the unused slots of every march_rays chunk (deltas == 0) are encoded as zeros
    without table gathers; composite_rays never reads them, so the rendered frame is bit-identical"""
    import s3d_hip
    from nerf import network_ff, synthetic as syn
    torch.manual_seed(0)
    net = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda().eval()
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    grid, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(grid))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    net.infer_batch_scale = 4
    poses = syn.orbit_poses(1, seed=0).cuda()
    r = syn.get_rays(poses, syn.lego_intrinsics(200, 200), 200, 200)
    ro, rd = r["rays_o"].contiguous(), r["rays_d"].contiguous()
    seen = []
    real = s3d_hip.active_live_rows

    def spy(B):
        t = real(B)
        if t is not None:
            seen.append(float((t[:, 0] == 0).float().mean()))
        return t
    monkeypatch.setattr(s3d_hip, "active_live_rows", spy)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = net.render(ro, rd, bg_color=1, perturb=False, max_steps=1024)
    assert seen and max(seen) > 0.05  # the mask was used and some chunks had dead slots
    monkeypatch.setattr(s3d_hip, "active_live_rows", lambda B: None)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        b = net.render(ro, rd, bg_color=1, perturb=False, max_steps=1024)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"])
    # direct: masked rows are exact zeros, live rows identical
    from tools.microbench import grid_meta
    offs, S, total = grid_meta()
    x = torch.rand(4096, 3, device="cuda")
    emb = (torch.rand(total, 2, device="cuda") - 0.5).half()
    deltas = torch.rand(4096, 2, device="cuda")
    deltas[::3] = 0
    o0 = torch.empty(16, 4096, 2, device="cuda", dtype=torch.half)
    o1 = torch.empty_like(o0)
    s3d_hip.GridBackend.grid_encode_forward(x, emb, offs, o0, 4096, 3, 2, 16, S, 16, None, 0, False, 0)
    s3d_hip.GridBackend.grid_encode_forward(x, emb, offs, o1, 4096, 3, 2, 16, S, 16, None, 0, False, 0, live=deltas)
    dead = deltas[:, 0] == 0
    assert not o1[:, dead].any() and torch.equal(o1[:, ~dead], o0[:, ~dead])
