"""GPU: the drop-in Python packages (autograd Functions / nn.Modules) on the product backend, compared with
the same wrappers driven by the CPU oracle — exercises the boundary exactly as the reference's callers do."""
import numpy as np
import pytest
import torch

from nerf import synthetic as syn

pytestmark = pytest.mark.gpu


def s3d_ctx(counter, rows):
    """the renderer's announcement of a device-side row count (s3d_hip.row_limit), or nothing"""
    import contextlib
    import s3d_hip
    return s3d_hip.row_limit(counter, rows) if counter is not None else contextlib.nullcontext()


def test_grid_encoder_module_autocast_and_grads(oracle, hip):
    import gridencoder.grid as gg
    torch.manual_seed(0)
    enc = gg.GridEncoder(desired_resolution=2048).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    x = (torch.rand(5000, 3, device="cuda") * 2 - 1)
    y = enc(x)
    assert y.shape == (5000, 32) and y.dtype == torch.float32
    (y * torch.linspace(0, 1, 32, device="cuda")).sum().backward()
    g32 = enc.embeddings.grad.clone()
    with torch.autocast("cuda", dtype=torch.float16):
        y16 = enc(x)
    assert y16.dtype == torch.float16
    torch.testing.assert_close(y16.float(), y, rtol=2e-2, atol=4e-3)
    # oracle-driven wrapper on the same weights
    ref = gg.GridEncoder(desired_resolution=2048)
    ref.embeddings.data.copy_(enc.embeddings.data.cpu())
    saved = gg._backend
    try:
        gg._backend = oracle.GridBackend
        yr = ref(x.cpu())
        (yr * torch.linspace(0, 1, 32)).sum().backward()
    finally:
        gg._backend = saved
    assert torch.equal(yr, y.detach().cpu()), "fp32 forward must be bit-exact vs the oracle"
    torch.testing.assert_close(g32.cpu(), ref.embeddings.grad, rtol=1e-5, atol=1e-5)


def test_raymarching_functions_end_to_end(oracle, hip):
    import raymarching.raymarching as rm
    grid, bits = syn.lego_like_density_grid(seed=0)
    poses = syn.orbit_poses(1, seed=1)
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=4096, generator=torch.Generator().manual_seed(0))
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    res = {}
    for dev in ("cpu", "cuda"):
        saved = rm._backend
        if dev == "cpu":
            rm._backend = oracle.RaymarchingBackend
        try:
            t = lambda v: v.to(dev)
            nears, fars = rm.near_far_from_aabb(t(ro), t(rd), t(aabb), 0.2)
            counter = torch.zeros(2, dtype=torch.int32, device=dev)
            xyzs, dirs, deltas, rays = rm.march_rays_train(t(ro), t(rd), 1.0, t(torch.from_numpy(bits)), 1, 128, nears, fars,
                                                          counter, -1, False, 128, False, 0, 1024)
            lo, hi = syn.lego_like_boxes(0)
            sig = (syn.box_density(xyzs.cpu(), lo, hi).to(dev) * 0.5).requires_grad_(True)
            rgb = (xyzs * 0.5 + 0.5).clamp(0, 1).detach().requires_grad_(True)
            ws, depth, image = rm.composite_rays_train(sig, rgb, deltas, rays, 1e-4)
            loss = (image * torch.tensor([1.0, 2.0, 3.0], device=dev)).sum() + ws.sum()
            loss.backward()
            res[dev] = [v.detach().cpu() for v in (xyzs, deltas, rays, counter, ws, depth, image, sig.grad, rgb.grad)]
            # packbits on the density grid reproduces the scene bitfield
            bf = rm.packbits(t(torch.from_numpy(grid)), 0.01)
            assert np.array_equal(bf.cpu().numpy(), bits)
        finally:
            rm._backend = saved
    a, b = res["cpu"], res["cuda"]
    assert a[0].shape == b[0].shape and a[0].shape[0] % 128 == 0
    for k in (0, 1, 2, 3):
        assert torch.equal(a[k], b[k])
    for k in (4, 5, 6, 8):
        torch.testing.assert_close(b[k], a[k], rtol=1e-4, atol=1e-6)
    assert ((b[7] - a[7]).abs().max() / a[7].abs().max()) < 1e-4


def test_sh_and_freq_modules(hip):
    from shencoder import SHEncoder
    from freqencoder import FreqEncoder
    d = torch.randn(1000, 3, device="cuda")
    d = d / d.norm(dim=-1, keepdim=True)
    sh = SHEncoder(degree=4)
    with torch.autocast("cuda", dtype=torch.float16):
        o = sh(d.half())
    assert o.dtype == torch.float32 and o.shape == (1000, 16)
    assert torch.allclose(o[:, 0], torch.full((1000,), 0.28209479177387814, device="cuda"))
    dd = d.clone().requires_grad_(True)
    sh(dd).sum().backward()
    assert dd.grad is not None and dd.grad.abs().sum() > 0
    fe = FreqEncoder(input_dim=3, degree=4)
    xx = (torch.rand(100, 3, device="cuda") - 0.5).requires_grad_(True)
    y = fe(xx)
    assert y.shape == (100, 27)
    y.sum().backward()
    ref = torch.ones_like(xx)
    for f in range(4):
        ref = ref + 2 ** f * (torch.cos(2 ** f * xx.detach()) - torch.sin(2 ** f * xx.detach()))
    torch.testing.assert_close(xx.grad, ref, rtol=1e-4, atol=1e-4)


def test_ffmlp_module(hip):
    from ffmlp import FFMLP
    net = FFMLP(32, 16, 64, 2).cuda()
    x = torch.randn(1000, 32, device="cuda") * 0.5
    net.train()
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    assert y.shape == (1000, 16) and y.dtype == torch.float16
    y.float().pow(2).sum().backward()
    assert net.weights.grad is not None and net.weights.grad.shape == net.weights.shape
    # torch twin
    w = net.weights.detach().half().float()
    m0, m1, m2 = w[:2048].view(64, 32), w[2048:2048 + 4096].view(64, 64), w[6144:].view(16, 64)
    xr = x.half().float()
    ref = torch.relu(torch.relu(xr @ m0.t()) @ m1.t()) @ m2.t()
    torch.testing.assert_close(y.float(), ref, rtol=3e-2, atol=3e-2)
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y2 = net(x)
    assert torch.equal(y2, y)


def test_network_ff_fused_head_matches_reference_op_sequence(hip):
    """nerf/network_ff.py forward/backward with the fused head glue (ngp_head.hip) vs the reference's torch op sequence
    (slice, trunc_exp, SH, cat, cast, sigmoid) on the same weights: same values up to fp16 rounding of identical
    intermediates, same gradients for the hash table and both MLPs."""
    from nerf import network_ff
    torch.manual_seed(0)
    net = network_ff.NeRFNetwork(bound=1, cuda_ray=True).cuda()
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    g = torch.Generator().manual_seed(1)
    B = 128 * 9
    x = (torch.rand(B, 3, generator=g) * 2 - 1).cuda()
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).cuda()
    w_s, w_c = torch.rand(B, generator=g).cuda(), torch.rand(B, 3, generator=g).cuda()
    res = {}
    for fused in (True, False):
        net.fused_head = fused
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb = net(x, d)
            loss = (sigma.float() * w_s).sum() * 1e-3 + (rgb.float() * w_c).sum()
        loss.backward()
        res[fused] = (sigma.float().detach().clone(), rgb.float().detach().clone(), net.encoder.embeddings.grad.clone(),
                      net.sigma_net.weights.grad.clone(), net.color_net.weights.grad.clone())
    net.fused_head = True
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-6, atol=0)      # exp(float(h0)): same expression
    torch.testing.assert_close(res[True][1], res[False][1], rtol=0, atol=1e-3)      # sigmoid rounded to fp16 both ways
    for i, name in ((2, "table"), (3, "sigma_net"), (4, "color_net")):
        a, b = res[True][i].float(), res[False][i].float()
        assert (a - b).norm() / b.norm().clamp(min=1e-12) < 2e-3, f"{name} gradient differs: {(a - b).norm() / b.norm()}"


@pytest.mark.parametrize("bound", [1, 2, 1.7])
def test_grid_encoder_fused_bound_normalisation(hip, bound):
    """GridEncoder.forward(x, bound) normalises inside the kernels when 2*bound is a power of two (else in torch); the result
    (and the table gradient) must be bit-identical to normalising in torch first, as gridencoder/grid.py:146 does."""
    from gridencoder import GridEncoder
    from gridencoder.grid import grid_encode
    torch.manual_seed(3)
    enc = GridEncoder(num_levels=8, base_resolution=8, log2_hashmap_size=14, desired_resolution=256).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    x = ((torch.rand(9000, 3) * 2 - 1) * bound * 1.01).cuda()  # a few points just outside
    g = torch.randn(9000, 16, device="cuda")
    outs = []
    for fused in (True, False):
        enc.zero_grad(set_to_none=True)
        if fused:
            y = enc(x, bound=bound)
        else:
            y = grid_encode((x + bound) / (2 * bound), enc.embeddings, enc.offsets, enc.per_level_scale, enc.base_resolution,
                            False, enc.gridtype_id, enc.align_corners, enc.interp_id)
        (y * g).sum().backward()
        outs.append((y.detach().clone(), enc.embeddings.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("autocast", [False, True])
def test_two_encoders_in_one_launch_match_two_calls(hip, autocast):
    """gridencoder.grid.grid_encode_pair (s3d_grid_encode_forward_pair: the density and the colour encoder of nerf/network.py on the
    same points) == the two GridEncoder calls: outputs and both table gradients bit for bit, with a device row count and a live mask"""
    from gridencoder import GridEncoder
    from gridencoder.grid import grid_encode_pair
    torch.manual_seed(9)
    a = GridEncoder(num_levels=16, base_resolution=16, log2_hashmap_size=15, desired_resolution=1024).cuda()
    b = GridEncoder(num_levels=16, base_resolution=16, log2_hashmap_size=15, desired_resolution=1024).cuda()
    a.embeddings.data.uniform_(-1, 1); b.embeddings.data.uniform_(-1, 1)
    B = 128 * 70
    x = ((torch.rand(B, 3) * 2 - 1) * 1.01).cuda()
    ga, gb = torch.randn(16, B, 2, device="cuda"), torch.randn(16, B, 2, device="cuda")
    nv = torch.tensor([B - 300], dtype=torch.int32, device="cuda")
    rows = (int(nv) + 127) // 128 * 128
    res = []
    for pair in (False, True):
        a.zero_grad(set_to_none=True); b.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            if pair:
                out = grid_encode_pair(a, b, x, 1, nv, None)
                assert out is not None
                ya, yb = out
            else:
                ya, yb = a(x, bound=1, level_major=True, n_valid=nv), b(x, bound=1, level_major=True, n_valid=nv)
        ((ya[:, :rows].float() * ga[:, :rows]).sum() + (yb[:, :rows].float() * gb[:, :rows]).sum()).backward()
        res.append((ya.detach()[:, :rows].clone(), yb.detach()[:, :rows].clone(), a.embeddings.grad.clone(), b.embeddings.grad.clone()))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    assert float(res[0][0].float().abs().max()) > 0 and float(res[0][3].abs().max()) > 0
    # inference: a live mask (rows whose first element is 0 come out as zeros)
    live = (torch.rand(B, 2, device="cuda") > 0.3).float()
    a.eval(); b.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
        ya, yb = grid_encode_pair(a, b, x, 1, None, live)
        za, zb = a(x, bound=1, level_major=True, live=live), b(x, bound=1, level_major=True, live=live)
    assert torch.equal(ya, za) and torch.equal(yb, zb)


def test_fused_background_mse_loss(hip):
    """nerf.trainer.render_loss with a deferred background == F.mse_loss(image + (1 - ws) * bg, gt), values and gradients"""
    from nerf.trainer import render_loss
    g = torch.Generator().manual_seed(7)
    N = 4096
    img = torch.rand(N, 3, generator=g).cuda().requires_grad_(True)
    ws = torch.rand(N, generator=g).cuda().requires_grad_(True)
    gt = torch.rand(N, 3, generator=g).cuda()
    for bg in (1, (0.2, 0.5, 0.9)):
        bgt = torch.tensor(bg if isinstance(bg, tuple) else (bg,) * 3, device="cuda", dtype=torch.float32)
        ref = torch.nn.functional.mse_loss(img + (1 - ws).unsqueeze(-1) * bgt, gt)
        gi_ref, gw_ref = torch.autograd.grad(ref * 1024.0, (img, ws))
        out = {"image": img, "weights_sum": ws, "premultiplied": True, "bg_color": bg}
        loss = render_loss(out, gt)
        gi, gw = torch.autograd.grad(loss * 1024.0, (img, ws))
        torch.testing.assert_close(loss, ref, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(gi, gi_ref, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(gw, gw_ref, rtol=1e-5, atol=1e-7)


def test_seal_network_fused_mlps_match_linear_op_sequence(hip):
    """nerf/network.py (the two-encoder net Seal-3D trains) under fp16 autocast: MFMA MLPs on weights packed from the nn.Linear
    parameters + the mid2 / rgb glue kernels vs the reference's op sequence (two GridEncoder calls, cat, hipBLASLt Linears,
    relu, trunc_exp, sigmoid) on the same parameters: same values up to fp16 rounding of identical intermediates, same
    gradients for both hash tables and all five Linear weights; density() agrees with forward()."""
    from nerf import network
    torch.manual_seed(0)
    net = network.NeRFNetwork(bound=1, cuda_ray=True, log2_hashmap_size=16).cuda()
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    net.encoder_color.embeddings.data.uniform_(-0.5, 0.5)
    g = torch.Generator().manual_seed(1)
    B = 128 * 9
    x = (torch.rand(B, 3, generator=g) * 2 - 1).cuda()
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).cuda()
    w_s, w_c = torch.rand(B, generator=g).cuda(), torch.rand(B, 3, generator=g).cuda()
    res = {}
    net.train()
    for fused in (True, False):
        net.fused_mlp = fused
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            assert net._can_fuse(x) == fused
            sigma, rgb = net(x, d)
            dens = net.density(x)["sigma"]
            loss = (sigma.float() * w_s).sum() * 1e-3 + (rgb.float() * w_c).sum()
        loss.backward()
        res[fused] = (sigma.float().detach().clone(), rgb.float().detach().clone(), dens.float().detach().clone(),
                      [p.grad.clone() for p in net.parameters()])
    net.fused_mlp = True
    # sigma = exp(h0) amplifies the fp16 rounding of h0 (|h0| up to a few units, half ulp 2^-10 relative): 4e-3
    torch.testing.assert_close(res[True][0], res[False][0], rtol=4e-3, atol=1e-6)
    torch.testing.assert_close(res[True][2], res[True][0], rtol=0, atol=0)          # density() == forward()'s sigma
    torch.testing.assert_close(res[True][1], res[False][1], rtol=0, atol=2e-3)      # sigmoid output rounded to fp16 both ways
    for (name, _), a, b in zip(net.named_parameters(), res[True][3], res[False][3]):
        a, b = a.float(), b.float()
        assert (a - b).norm() / b.norm().clamp(min=1e-12) < 5e-3, f"{name} gradient differs: {(a - b).norm() / b.norm()}"
    # inference mode (no stored activations) gives the training-mode values
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s2, c2 = net(x, d)
    assert torch.equal(s2.float(), res[True][0]) and torch.equal(c2.float(), res[True][1])


@pytest.mark.parametrize("B,ragged", [(128 * 9, 0), (128 * 40, 333)])
def test_seal_network_one_launch_pair_matches_the_three_launches(hip, B, ragged):
    """nerf/network.py's fused path with the one-launch MLP pair (k_ffmlp_ngp_pair<true>: density MLP, trunc_exp, colour-net input
    row with the second encoder's features, colour MLP, sigmoid) against the same path as three launches (density MLP, mid2
    kernel, colour MLP): sigma bit for bit, rgb equal except where an inlined SH term rounds the other way (one fp16 ulp of a
    colour-net input), every gradient to fp16 accuracy; with a device-side row limit the rows behind it are never read back."""
    from nerf import network
    torch.manual_seed(3)
    net = network.NeRFNetwork(bound=1, cuda_ray=True, log2_hashmap_size=16).cuda()
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    net.encoder_color.embeddings.data.uniform_(-0.5, 0.5)
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(B, 3, generator=g) * 2 - 1).cuda()
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).cuda()
    w_s, w_c = torch.rand(B, generator=g).cuda(), torch.rand(B, 3, generator=g).cuda()
    rows = B
    if ragged:
        nv = torch.tensor([B - ragged], dtype=torch.int32, device="cuda")
        rows = B - ragged
    res = {}
    net.train()
    for pair in (True, False):
        net.fused_pair = pair
        net.zero_grad(set_to_none=True)
        with s3d_ctx(nv if ragged else None, B), torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb = net(x, d)
            loss = (sigma.float()[:rows] * w_s[:rows]).sum() * 1e-3 + (rgb.float()[:rows] * w_c[:rows]).sum()
        loss.backward()
        res[pair] = (sigma.float().detach()[:rows].clone(), rgb.float().detach()[:rows].clone(),
                     [p.grad.clone() for p in net.parameters()])
    net.fused_pair = True
    assert torch.equal(res[True][0], res[False][0])
    assert float((res[True][1] != res[False][1]).any(-1).float().mean()) < 2e-3
    torch.testing.assert_close(res[True][1], res[False][1], rtol=0, atol=2e-3)
    for (name, _), a, b in zip(net.named_parameters(), res[True][2], res[False][2]):
        a, b = a.float(), b.float()
        assert float(b.abs().max()) > 0, name
        assert (a - b).norm() / b.norm() < 5e-3, f"{name} gradient differs: {(a - b).norm() / b.norm()}"
    net.eval()
    with torch.no_grad(), s3d_ctx(nv if ragged else None, B), torch.autocast("cuda", dtype=torch.float16):
        s2, c2 = net(x, d)
    assert torch.equal(s2.float()[:rows], res[True][0]) and torch.equal(c2.float()[:rows], res[True][1])


def test_cached_half_weights_follow_a_fused_torch_optimizer(hip):
    """Adam(fused=True) updates parameters in place WITHOUT bumping Tensor._version; the no-grad / eval-mode caches of
    fp16 casts (grid table, FFMLP weights) must still see every step (they key on the optimizer-step epoch)."""
    import gridencoder.grid as gg
    from ffmlp import FFMLP
    torch.manual_seed(0)
    enc = gg.GridEncoder(desired_resolution=256, log2_hashmap_size=14).cuda()
    enc.embeddings.data.uniform_(-1, 1)
    mlp = FFMLP(32, 16, 64, 2).cuda()
    opt = torch.optim.Adam(list(enc.parameters()) + list(mlp.parameters()), lr=1e-1, fused=True)
    scaler = torch.amp.GradScaler("cuda")
    x = torch.rand(4096, 3, device="cuda") * 2 - 1

    def infer():
        enc.eval(), mlp.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            h = enc(x)
            return h.float().clone(), mlp(h).float().clone()

    e0, m0 = infer()
    e0b, m0b = infer()  # second call: served from the cache
    assert torch.equal(e0, e0b) and torch.equal(m0, m0b)
    for _ in range(3):
        enc.train(), mlp.train()
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            loss = (mlp(enc(x)).float() ** 2).mean()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
    e1, m1 = infer()
    assert not torch.equal(e0, e1), "eval-mode grid output is stale after optimizer steps"
    assert not torch.equal(m0, m1), "no-grad FFMLP output is stale after optimizer steps"
    # and equals what freshly cast weights give
    for p in list(enc.parameters()) + list(mlp.parameters()):  # (the cached copies live on the parameters)
        if hasattr(p, "_s3d_eval_half"):
            del p._s3d_eval_half
    e2, m2 = infer()
    assert torch.equal(e1, e2) and torch.equal(m1, m2)
    # writes through `.data` bump no version: reset_parameters() announces them itself
    mlp.reset_parameters()
    enc.reset_parameters()
    e3, m3 = infer()
    assert not torch.equal(e2, e3) and not torch.equal(m2, m3), "stale fp16 copies after reset_parameters()"
    assert enc.embeddings._s3d_eval_half[2].data_ptr() != 0 and not hasattr(gg, "_half_cache")  # freed with the parameter


def test_packed_linear_weights_in_the_native_optimizer_match_the_assembled_path(hip):
    """nerf/network.py: PackedWeights.  With NativeAdam the five nn.Linear weights of the Seal network live as row-strided
    views of two packed fp16 buffers (weights + gradient twin): the MLP backward writes the packed gradient, the Adam launch
    reads / writes through the views.  Against the same model whose packs are switched off (weights assembled by cat / pad per
    call, fp16 weight gradient split back into fp32 `.grad`s by autograd): identical parameters after three steps — the fp32
    `.grad`s are exact images of the fp16 values, the update arithmetic is the same — and the pack holds the fp16 image of the
    assembled weights."""
    import bench
    import s3d_hip
    from nerf import network, synthetic as syn
    from nerf.trainer import Trainer
    grid, bits = syn.lego_like_density_grid(seed=0)
    batches, _ = bench.make_batches(3, 1024, 0, torch.device("cuda"), s3d_hip.RaymarchingBackend, torch.from_numpy(bits).cuda(),
                                    syn.lego_like_boxes(0))
    out = {}
    for packed in (True, False):
        torch.manual_seed(0)
        net = network.NeRFNetwork(bound=1, cuda_ray=True, log2_hashmap_size=15, density_scale=1, min_near=0.2, density_thresh=10).cuda()
        net.density_grid.copy_(torch.from_numpy(grid))
        net.density_bitfield.copy_(torch.from_numpy(bits))
        net.iter_density = 100
        if not packed:
            for pk in net._packs:
                for p, _, _ in pk.members:
                    del p._s3d_pack_spec
            net._packs = None
        tr = Trainer(net, lr=1e-2, fp16=True, update_extra_interval=10 ** 9)
        tr.global_step = 1
        assert tr.native_optim and (net._packs is None or all(pk.adopted for pk in net._packs))
        losses = []
        for i in range(3):
            torch.manual_seed(100 + i)  # (the marcher's jitter)
            losses.append(float(tr.train_step(*batches[i])))
        if packed:
            for pk in net._packs:
                assert pk.current() is not None
                for p, _, _ in pk.members:
                    assert p.grad is None and p._s3d_grad_consumed  # the gradient never became a `.grad`
            ws, wc = net._packed_weights()
            assert torch.equal(net._packs[0].half, ws.half()) and torch.equal(net._packs[1].half, wc.half())
        out[packed] = (losses, {k: v.detach().clone() for k, v in net.named_parameters()})
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    for k, v in out[True][1].items():
        assert torch.equal(v, out[False][1][k]), k


def test_two_backward_passes_of_one_step_add_up_in_the_packed_weight_gradient(hip):
    """`NeRFRenderer.run` (cuda_ray off, nerf/renderer.py:168-214) queries `density()` twice with gradients before one
    optimizer step, and gradient accumulation runs several backward passes per step.  The pack's gradient twin is never cleared,
    so the FIRST backward since zero_grad() / step() replaces it and every further one adds to it (PackedWeights._s3d_overwrite):
    the twin must hold the SUM of both passes — against the same model without packs, where autograd sums the fp32 `.grad`s —
    and the next step must start from a replaced twin again."""
    from nerf import network
    from nerf.trainer import Trainer
    g = torch.Generator().manual_seed(5)
    xs = [((torch.rand(1024, 3, generator=g) * 2 - 1) * 0.9).cuda() for _ in range(3)]
    res = {}
    for packed in (True, False):
        torch.manual_seed(0)
        net = network.NeRFNetwork(bound=1, cuda_ray=True, log2_hashmap_size=15, density_scale=1, min_near=0.2, density_thresh=10).cuda()
        with torch.no_grad():
            net.encoder.embeddings.uniform_(-0.5, 0.5)
        if not packed:
            for pk in net._packs:
                for p, _, _ in pk.members:
                    del p._s3d_pack_spec
            net._packs = None
        tr = Trainer(net, lr=1e-2, fp16=True, update_extra_interval=10 ** 9)
        assert tr.native_optim
        net.train()
        out = []
        for batches in ((xs[0], xs[1]), (xs[2],)):       # step A: two passes; step B: one pass (must not see step A's sum)
            tr.optimizer.zero_grad()
            for x in batches:
                with torch.autocast("cuda", dtype=torch.float16):
                    sig = net.density(x)["sigma"]
                (sig.float().sum() * 4.0).backward()
            ws = [l.weight for l in net.sigma_net]
            if packed:
                assert all(w.grad is None and w._s3d_grad_touched for w in ws)
                out.append([w._s3d_grad.float().clone() for w in ws])
            else:
                out.append([w.grad.float().clone() for w in ws])
            if batches is not None and len(batches) == 2:
                for w in ws:      # as a consuming step would leave the flags (the twin itself stays as it is: never cleared)
                    if packed:
                        w._s3d_grad_consumed = True
        res[packed] = out
    for step in range(2):
        for a, b in zip(res[True][step], res[False][step]):
            assert b.abs().max() > 0
            torch.testing.assert_close(a, b, rtol=1e-2, atol=2e-3 * float(b.abs().max()))
    # the two-pass sum really is larger than one pass (the bug this guards against kept only the second pass)
    assert not torch.allclose(res[True][0][0], res[True][1][0], rtol=0.2, atol=0.0)


def test_seal_loss_heads_match_the_torch_op_sequences(hip):
    """The Seal-3D loss glue as single launches (csrc/ngp_head.hip) against the reference's op sequences: fine-tuning criterion
    MSE + L1(nan_to_num(depth)) (nerf/utils.py:484-489; the depth term has a value and no gradient, raymarching.py:274),
    pretraining criterion L1(sigma) + L1(colour) with padding rows (SealNeRF/trainer.py:455-469), teacher targets
    nan_to_num(image + (1 - ws) * bg) / nan_to_num(depth) (SealNeRF/trainer.py:572-582)."""
    from nerf.trainer import render_loss
    from sealnerf.trainer import _L1Pair
    g = torch.Generator().manual_seed(3)
    N = 4096
    img = torch.rand(N, 3, generator=g).cuda().requires_grad_(True)
    ws = torch.rand(N, generator=g).cuda().requires_grad_(True)
    gt = torch.rand(N, 3, generator=g).cuda()
    depth = (torch.rand(N, generator=g) * 3).cuda()
    depth[::97] = float("nan")
    gtd = (torch.rand(N, generator=g) * 3).cuda()
    scale = torch.tensor(512.0, device="cuda")
    ref = torch.nn.functional.mse_loss(img + (1 - ws).unsqueeze(-1) * 1.0, gt) + 0.7 * torch.nn.functional.l1_loss(torch.nan_to_num(depth, nan=0.0), gtd)
    gi_ref, gw_ref = torch.autograd.grad(ref * scale, (img, ws))
    out = {"image": img, "weights_sum": ws, "depth": depth, "premultiplied": True, "bg_color": 1}
    for expected in (None, scale):
        loss = render_loss(out, gt, expected, gtd, 0.7)
        gi, gw = torch.autograd.grad(loss, (img, ws), grad_outputs=scale)
        torch.testing.assert_close(loss, ref, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(gi, gi_ref, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(gw, gw_ref, rtol=1e-5, atol=1e-7)
    # pretraining criterion, 300,000 points + 37 padding rows, shard normalisation
    n, pad, n_total = 300000, 37, 450000
    sig = (torch.rand(n + pad, generator=g) * 4).cuda().requires_grad_(True)
    col = torch.rand(n + pad, 3, generator=g).cuda().requires_grad_(True)
    gs, gc = (torch.rand(n, generator=g) * 4).cuda(), torch.rand(n, 3, generator=g).cuda()
    with torch.no_grad():
        sig[5], col[7, 1] = gs[5], gc[7, 1]  # exact ties: sign(0) = 0
    ref = (sig[:n] - gs).abs().sum() / n_total + (col[:n] - gc).abs().sum() / (3 * n_total)
    rs, rc = torch.autograd.grad(ref * scale, (sig, col))
    vals = []
    for expected in (None, scale):
        loss = _L1Pair.apply(sig, col, gs, gc, n_total, expected)
        a, b = torch.autograd.grad(loss, (sig, col), grad_outputs=scale)
        torch.testing.assert_close(loss, ref, rtol=2e-5, atol=1e-7)
        torch.testing.assert_close(a, rs, rtol=1e-6, atol=0)
        torch.testing.assert_close(b, rc, rtol=1e-6, atol=0)
        assert float(a[n:].abs().max()) == 0.0 and float(b[n:].abs().max()) == 0.0 and float(a[5]) == 0.0 and float(b[7, 1]) == 0.0
        vals.append(float(loss))
    assert vals[0] == vals[1] == float(_L1Pair.apply(sig, col, gs, gc, n_total, None))  # fixed summation order
    # teacher targets
    image = torch.rand(N, 3, generator=g).cuda()
    image[3, 1] = float("nan")
    w = torch.rand(N, generator=g).cuda()
    dep = depth.clone()
    dep[11] = float("inf")
    o_rgb, o_dep = torch.empty(N, 3, device="cuda"), torch.empty(N, device="cuda")
    hip.NgpHeadBackend.bg_targets(image, w, dep, (1.0, 1.0, 1.0), o_rgb, o_dep)
    assert torch.equal(o_rgb, torch.nan_to_num(image + (1 - w).unsqueeze(-1) * 1, nan=0.0))
    assert torch.equal(o_dep, torch.nan_to_num(dep, nan=0.0))
