"""GPU: nerf.optim.NativeAdam / NativeGradScaler against torch.optim.Adam + torch.amp.GradScaler (the reference's
update, nerf/utils.py:356-361), including the fp16 gradient hand-over of GridEncoder and the skipped step on overflow."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def test_native_adam_matches_torch_adam(hip):
    from nerf.optim import NativeAdam
    p0 = [_mk((1000, 2), 1), _mk((4097,), 2)]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    oa = NativeAdam([{"params": pa}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    scale = torch.full((1,), 128.0, device="cuda")
    flag = torch.zeros(1, device="cuda")
    for step in range(6):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = _mk(a.shape, 10 * step + i).cuda() * (10.0 ** (i - 1))
            g[::7] = 0  # exact zeros keep m/v decaying only
            a.grad = g * 128.0  # scaled gradient, as GradScaler hands it over
            b.grad = g.clone()
        flag.fill_(1.0 if step == 3 else 0.0)  # step 3 is an overflow step: must be skipped entirely
        oa.step(grad_scale=scale, found_inf=flag)
        if step != 3:
            ob.step()
    assert float(oa.step_count) == 5.0
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-6, atol=1e-7)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-12)


def test_adam_multi_tensor_launch_is_the_per_tensor_update_bit_for_bit(hip):
    """s3d_adam_step_multi == s3d_adam_step per tensor: fp16 and fp32 gradients, sizes with vector tails, a tensor whose
    pointers are not 16-byte aligned (scalar path), fp16 parameter copies; a raised flag skips every tensor."""
    import s3d_hip
    O = s3d_hip.OptimBackend
    torch.manual_seed(3)
    sizes = [12_000_003, 11264, 7168, 5, 4099] + [0, 33, 64] * 5  # 20 tensors: more than one launch's worth, some empty
    step = torch.full((1,), 4.0, device="cuda")
    scale = torch.full((1,), 1024.0, device="cuda")
    flag = torch.zeros(1, device="cuda")

    def make():
        out = []
        for k, n in enumerate(sizes):
            off = 1 if k == 4 else 0  # tensor 4: views shifted by one element -> unaligned
            p = torch.randn(n + off, device="cuda")[off:]
            g = (torch.randn(n + off, device="cuda") * 1024.0)
            g = (g.half() if k < 3 else g)[off:]
            m = (torch.randn(n + off, device="cuda") * 0.1)[off:]
            v = (torch.rand(n + off, device="cuda") * 0.01)[off:]
            h = torch.empty(n + off, device="cuda", dtype=torch.half)[off:] if k in (0, 1, 4, 7, 12) else None
            out.append([p, g, m, v, h])
        return out
    torch.manual_seed(3); a = make()
    torch.manual_seed(3); b = make()
    for x, y in zip(a, b):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])
    for p, g, m, v, h in a:
        if p.numel():
            O.adam_step(p, g, m, v, h, 1e-2, 0.9, 0.99, 1e-15, step, scale, flag)
    O.adam_step_multi([(p, g, m, v, h, 1e-2, 0.9, 0.99, 1e-15) for p, g, m, v, h in b], step, scale, flag)
    for x, y in zip(a, b):
        for k in (0, 2, 3):
            assert torch.equal(x[k], y[k])
        if x[4] is not None:
            assert torch.equal(x[4], y[4]) and torch.equal(x[4], x[0].half())
    before = [t[0].clone() for t in b]
    flag.fill_(1.0)
    O.adam_step_multi([(p, g, m, v, h, 1e-2, 0.9, 0.99, 1e-15) for p, g, m, v, h in b], step, scale, flag)
    assert all(torch.equal(x, t[0]) for x, t in zip(before, b))


def test_scaler_update_carries_the_step_count_advance(hip):
    """NativeGradScaler.step + update: the optimizer's step count advances exactly once per clean step (folded into update's
    launch) and not at all on an overflow step"""
    from nerf.optim import NativeAdam, NativeGradScaler
    p = torch.nn.Parameter(torch.randn(1000, device="cuda"))
    opt = NativeAdam([{"params": [p]}], lr=1e-2)
    sc = NativeGradScaler(p.device, enabled=True)
    for k in range(4):
        p.grad = torch.randn(1000, device="cuda") * float(sc._scale)
        if k == 2:
            p.grad[5] = float("inf")
        before = p.detach().clone()
        sc.step(opt)
        sc.update()
        assert float(opt.step_count) == (k + 1 if k < 2 else k)
        assert torch.equal(before, p.detach()) == (k == 2)
    assert float(sc._found_inf) == 0.0 and float(sc._scale) == 2.0 ** 15


def test_grid_encoder_half_grad_handover_matches_reference_update(hip):
    """Same data through (a) autograd fp32 grads + torch Adam + GradScaler and (b) fp16 hand-over + NativeAdam +
    NativeGradScaler: the tables must follow the same trajectory (the fp16 gradient values are identical, only their
    route to the optimizer differs), and the fp16 copy used by the next forward must equal the cast of the table."""
    from gridencoder import GridEncoder
    from nerf.optim import NativeAdam, NativeGradScaler
    torch.manual_seed(0)
    ea = GridEncoder(num_levels=4, base_resolution=4, log2_hashmap_size=10, desired_resolution=32).cuda()
    eb = GridEncoder(num_levels=4, base_resolution=4, log2_hashmap_size=10, desired_resolution=32).cuda()
    eb.load_state_dict(ea.state_dict())
    w = _mk((8, 1), 3).cuda()
    oa = NativeAdam([{"params": ea.parameters()}], lr=1e-2)
    sa = NativeGradScaler("cuda", init_scale=1024.0)
    ob = torch.optim.Adam(eb.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sb = torch.amp.GradScaler("cuda", init_scale=1024.0)
    assert ea.embeddings._s3d_grad.dtype == torch.float16
    for step in range(4):
        x = (torch.rand(9000, 3, generator=torch.Generator().manual_seed(step)) * 2 - 1).cuda()
        for enc, opt, sc in ((ea, oa, sa), (eb, ob, sb)):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                y = enc(x, bound=1)
                loss = ((y.float() @ w) ** 2).mean() * (1e8 if step == 2 else 1.0)  # step 2 overflows fp16 -> skipped
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
        assert ea.embeddings.grad is None, "the table gradient must not go through autograd"
        # step 0 agrees to an fp32 ulp; afterwards an ulp of difference in the table can flip a bit of its fp16 copy,
        # and Adam with eps = 1e-15 turns gradient noise of rarely-hit rows (v ~ 1e-18) into percent-level changes of
        # their 1e-2-sized updates: trajectories are compared at 1e-3 of an update
        torch.testing.assert_close(ea.embeddings.detach(), eb.embeddings.detach(), rtol=1e-3, atol=(1e-8 if step == 0 else 2e-5))
        assert torch.equal(ea.embeddings._s3d_half, ea.embeddings.detach().half())
        assert sa.get_scale() == sb.get_scale()
    assert sa.get_scale() == 512.0  # one overflow halved the scale
    # writing the parameter through torch invalidates the fp16 copy; the next forward must see the new values
    with torch.no_grad():
        ea.embeddings.mul_(2.0)
    x = (torch.rand(8192, 3) * 2 - 1).cuda()
    with torch.autocast("cuda", dtype=torch.float16):
        y2 = ea(x, bound=1)
        eb.embeddings.data.copy_(ea.embeddings.data)
        y_ref = eb(x, bound=1)
    assert torch.equal(y2, y_ref)


def test_native_scaler_growth(hip):
    from nerf.optim import NativeAdam, NativeGradScaler
    p = torch.nn.Parameter(torch.ones(64, device="cuda"))
    opt = NativeAdam([{"params": [p]}], lr=1e-3)
    sc = NativeGradScaler("cuda", init_scale=4.0, growth_interval=2)
    for i in range(4):
        p.grad = torch.ones_like(p)
        sc.step(opt)
        sc.update()
    assert sc.get_scale() == 16.0 and float(opt.step_count) == 4.0
    p.grad = torch.full_like(p, float("nan"))
    before = p.detach().clone()
    sc.step(opt)
    sc.update()
    assert sc.get_scale() == 8.0 and float(opt.step_count) == 4.0 and torch.equal(before, p.detach())


def test_ffmlp_weight_grad_handover(hip):
    """FFMLP weights adopted by NativeAdam: the gradient lands in the optimizer's flat fp16 buffer (not in `.grad`), with
    the values the autograd route produces, and accumulates over two backward passes of one step."""
    from ffmlp import FFMLP
    from nerf.optim import NativeAdam
    torch.manual_seed(1)
    ma, mb = FFMLP(32, 16, 64, 2).cuda(), FFMLP(32, 16, 64, 2).cuda()
    mb.load_state_dict(ma.state_dict())
    opt = NativeAdam([{"params": ma.parameters()}], lr=1e-3)
    assert opt.flat_half is not None and ma.weights._s3d_grad_flat is opt.flat_half
    x = torch.randn(1024, 32, device="cuda")
    opt.zero_grad()
    for m in (ma, mb):
        for _ in range(2):
            with torch.autocast("cuda", dtype=torch.float16):
                (m(x).float() ** 2).mean().backward()
    assert ma.weights.grad is None and ma.weights._s3d_grad_touched
    torch.testing.assert_close(ma.weights._s3d_grad.float(), mb.weights.grad, rtol=2e-3, atol=1e-4)


def _through_file(obj):
    import io
    buf = io.BytesIO()
    torch.save(obj, buf)
    buf.seek(0)
    return torch.load(buf, weights_only=False)


def test_optimizer_and_scaler_state_dicts_interchange_with_torch(hip):
    """a `full` checkpoint (nerf/utils.py:1031-1036 of the reference) carries torch.optim.Adam / GradScaler state: state
    saved by the native pair resumes under torch's and the other way round, and both continue identically"""
    from nerf.optim import NativeAdam, NativeGradScaler
    p0 = [_mk((513, 2), 1), _mk((777,), 2)]

    def grads(params, step):
        for i, p in enumerate(params):
            p.grad = _mk(p.shape, 100 + 10 * step + i).cuda()

    pa = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    oa = NativeAdam([{"params": pa}], lr=1e-2)
    for step in range(3):
        grads(pa, step)
        oa.step()
    sd = oa.state_dict()
    assert all(float(st["step"]) == 3.0 and set(st) == {"step", "exp_avg", "exp_avg_sq"} for st in sd["state"].values())

    pb = [torch.nn.Parameter(a.detach().clone()) for a in pa]
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob.load_state_dict(_through_file(sd))  # (load_state_dict keeps same-dtype tensors by reference: go through a file)
    pc = [torch.nn.Parameter(a.detach().clone()) for a in pa]
    oc = NativeAdam([{"params": pc}], lr=1e-2)
    oc.load_state_dict(_through_file(ob.state_dict()))  # torch -> native
    assert float(oc.step_count) == 3.0
    for step in range(3, 6):
        for o, ps in ((oa, pa), (ob, pb), (oc, pc)):
            grads(ps, step)
            o.step()
    for a, b, c in zip(pa, pb, pc):
        assert torch.equal(a, c)  # native resumed == native uninterrupted, bit for bit
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-6, atol=1e-7)

    sa = NativeGradScaler("cuda")
    sa._scale.fill_(4096.0)
    sa._growth_tracker.fill_(17)
    sb = torch.amp.GradScaler("cuda")
    sb.load_state_dict(sa.state_dict())
    assert sb.state_dict() == sa.state_dict()
    sc = NativeGradScaler("cuda")
    sc.load_state_dict(sb.state_dict())
    assert sc.get_scale() == 4096.0 and int(sc._growth_tracker) == 17


def test_lr_schedule_reaches_a_captured_step_through_the_device_side_factor(hip):
    """A learning-rate schedule (LambdaLR in main_SealNeRF.py:283-288) moves `param_groups[i]["lr"]` every step.  A step
    captured in a HIP graph has the lr of the capture among its launch arguments; `NativeAdam.capture_lr` /
    `follow_lr_schedule` route the later values through one device word (s3d_adam_step_multi: lr_scale).  The replayed
    updates must equal torch.optim.Adam under the same schedule."""
    from nerf.optim import NativeAdam
    p0 = [_mk((3000, 2), 1), _mk((513,), 2)]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    oa = NativeAdam([{"params": pa[:1]}, {"params": pa[1:], "lr": 3e-3}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam([{"params": pb[:1]}, {"params": pb[1:], "lr": 3e-3}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    base = [g["lr"] for g in oa.param_groups]
    sg = [torch.zeros_like(p) for p in pa]  # static gradient buffers of the captured step
    for p, g in zip(pa, sg):
        p.grad = g
    steps = 12
    grads = [[_mk(p.shape, 100 * s + i).cuda() for i, p in enumerate(pa)] for s in range(steps)]
    for g, src in zip(sg, grads[0]):
        g.copy_(src)
    oa.capture_lr()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        oa.step()  # warm-up = step 0 (eager, factor 1)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        oa.step()
    for s in range(steps):
        f = 0.1 ** (s / steps)
        for grp, grb, b in zip(oa.param_groups, ob.param_groups, base):
            grp["lr"] = grb["lr"] = b * f
        for i, b in enumerate(pb):
            b.grad = grads[s][i].clone()
        ob.step()
        if s > 0:
            for g, src in zip(sg, grads[s]):
                g.copy_(src)
            assert oa.follow_lr_schedule()
            graph.replay()
    assert float(oa.step_count) == steps
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=3e-6, atol=2e-7)
    # groups that move apart cannot share the factor: the caller is told to re-capture
    oa.param_groups[0]["lr"] *= 0.5
    assert not oa.follow_lr_schedule()
    # ... and an eager step rebases by itself
    for g, src in zip(sg, grads[0]):
        g.copy_(src)
    before = pa[0].detach().clone()
    oa.step()
    assert oa._lr_captured[0] == oa.param_groups[0]["lr"] and float(oa.lr_scale) == 1.0
    assert not torch.equal(before, pa[0].detach())


def test_native_adam_forms_an_announced_l1_gradient_inside_the_update(hip):
    """s3d_adam_tensor.l1 (tensoRF/utils.py: TensoRF's L1 penalty on the density factors): p._s3d_l1 = c announced for ONE step
    adds c * sign(p) to the unscaled gradient — against torch.optim.Adam on grad + c * torch.sign(p); the announcement is consumed
    by the step, exact zeros of the parameter take no penalty gradient, an overflow step applies nothing"""
    from nerf.optim import NativeAdam
    p0 = [_mk((600, 5), 1), _mk((4097,), 2)]
    p0[1][::11] = 0.0
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in p0]
    oa = NativeAdam([{"params": pa}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    scale = torch.full((1,), 256.0, device="cuda")
    flag = torch.zeros(1, device="cuda")
    coef = [1e-4 / p0[0].numel(), 3e-2]
    for step in range(5):
        announced = step != 2  # (step 2: no announcement -> a plain update)
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = _mk(a.shape, 10 * step + i).cuda() * 1e-3
            g[::5] = 0
            a.grad = g * 256.0
            b.grad = g + (coef[i] * torch.sign(b.detach()) if announced and step != 3 else 0)
            if announced:
                a._s3d_l1 = coef[i]
        flag.fill_(1.0 if step == 3 else 0.0)
        oa.step(grad_scale=scale, found_inf=flag)
        assert all("_s3d_l1" not in a.__dict__ for a in pa)
        if step != 3:
            ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-6, atol=2e-7)
    assert torch.equal(pa[1].detach()[::11][:1] == 0, pb[1].detach()[::11][:1] == 0)


def test_grid_backward_with_adam_inside_matches_backward_then_adam(hip):
    """s3d_grid_encode_backward_adam vs s3d_grid_encode_backward + s3d_adam_step_multi on the same inputs, two consecutive
    steps (non-zero moments on the second), loss scale 1024: master table, moments and fp16 copy bit for bit; rows no point
    touches move on their moments alone; a raised flag or a non-finite dL/dy leaves everything as it was (and raises the flag)."""
    import numpy as np
    import s3d_hip
    from gridencoder import GridEncoder
    G, O = s3d_hip.GridBackend, s3d_hip.OptimBackend
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)
    offs = enc.offsets.cuda()
    rows = int(offs[-1])
    L, S = 16, float(np.log2(enc.per_level_scale))
    B = 1 << 16
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, generator=gen).cuda()
    x[: B // 2] = x[: B // 2] * 0.25 + 0.3  # half of the points in a corner of the volume: most fine-level rows stay untouched
    scale = torch.full((1,), 1024.0, device="cuda")
    step = torch.zeros(1, device="cuda")

    def state(seed):
        g = torch.Generator().manual_seed(seed)
        p = (torch.rand(rows, 2, generator=g) * 2e-4 - 1e-4).cuda()
        return [p, torch.zeros_like(p), torch.zeros_like(p), p.half()]
    a, b = state(1), state(1)
    table = a[3].clone()
    for it in range(2):
        grad = (torch.randn(L, B, 2, generator=gen) * 3.0).half().cuda()
        flag = torch.zeros(1, device="cuda")
        ge = torch.zeros(rows, 2, dtype=torch.half, device="cuda")
        G.grid_encode_backward(grad, x, table, offs, ge, B, 3, 2, L, S, 16, None, None, 0, False, 0, found_inf=flag)
        assert float(flag) == 0.0
        O.adam_step_multi([(a[0], ge, a[1], a[2], a[3], 1e-2, 0.9, 0.99, 1e-15)], step, scale, flag)
        ge2 = torch.zeros(rows, 2, dtype=torch.half, device="cuda")
        adam = dict(param=b[0], exp_avg=b[1], exp_avg_sq=b[2], param_half=b[3], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, step=step,
                    grad_scale=scale, lr_scale=None)
        assert G.grid_encode_backward_adam(grad, x, table, offs, ge2, B, 3, 2, L, S, 16, 0, False, 0, adam, found_inf=flag)
        assert float(ge2.abs().max()) == 0.0 and float(flag) == 0.0
        for k in range(4):
            assert torch.equal(a[k], b[k]), (it, k, float((a[k].float() - b[k].float()).abs().max()))
        if it == 1:
            untouched = (ge == 0).all(1)
            assert 0.2 < float(untouched.float().mean()) < 0.99
        step += 1
    before = [t.clone() for t in b]
    flag = torch.ones(1, device="cuda")  # another producer of the step already raised the flag: skipped as a whole
    assert G.grid_encode_backward_adam(grad, x, table, offs, ge2, B, 3, 2, L, S, 16, 0, False, 0, adam, found_inf=flag)
    assert all(torch.equal(u, w) for u, w in zip(before, b))
    flag = torch.zeros(1, device="cuda")
    bad = grad.clone()
    bad[7, 12345, 1] = float("inf")
    assert G.grid_encode_backward_adam(bad, x, table, offs, ge2, B, 3, 2, L, S, 16, 0, False, 0, adam, found_inf=flag)
    assert float(flag) == 1.0 and all(torch.equal(u, w) for u, w in zip(before, b))
    # the next clean call finds clean control words
    flag = torch.zeros(1, device="cuda")
    assert G.grid_encode_backward_adam(grad, x, table, offs, ge2, B, 3, 2, L, S, 16, 0, False, 0, adam, found_inf=flag)
    assert float(flag) == 0.0 and not torch.equal(before[0], b[0])
    # small batches take the direct-atomics kernels: not applied, the gradient is in the table
    small = G.grid_encode_backward_adam(grad[:, :4096].contiguous(), x[:4096].contiguous(), table, offs, ge2, 4096, 3, 2, L, S, 16, 0, False,
                                        0, adam, found_inf=flag)
    assert small is False and float(ge2.abs().max()) > 0
