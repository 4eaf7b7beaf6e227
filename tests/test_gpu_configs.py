"""GPU smoke/parity of the remaining BASELINE configurations on the product path:
config 3 (Seal bbox distillation, teacher+student NGP) and config 5 (TensoRF VM backbone with the HIP freq encoder)."""
import numpy as np
import pytest
import torch

from nerf import synthetic as syn

pytestmark = pytest.mark.gpu

BBOX = {"type": "bbox",
        "raw": [[x, y, z] for x in (-0.2, 0.2) for y in (0.0, 0.3) for z in (-0.2, 0.2)],
        "transform": [[1, 0, 0, 0.3], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], "scale": [1, 1, 1], "boundType": "both"}


def test_seal_bbox_distillation_on_gpu(hip):
    from nerf import network
    from sealnerf import SealBBoxMapper, SealTrainer, make_student, make_teacher
    torch.manual_seed(0)
    kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    teacher = make_teacher(network.NeRFNetwork, **kw).cuda()
    student = make_student(network.NeRFNetwork, **kw).cuda()
    # a "pretrained" teacher: occupancy of the synthetic scene + some structure in the weights
    grid, bits = syn.lego_like_density_grid(seed=0)
    teacher.density_grid.copy_(torch.from_numpy(grid))
    teacher.density_bitfield.copy_(torch.from_numpy(bits))
    for p in teacher.parameters():
        p.data.uniform_(-0.2, 0.2)
    student.load_state_dict(teacher.state_dict())
    mapper = SealBBoxMapper(BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    tr = SealTrainer(student, teacher, lr=1e-2, fp16=True)
    # BASELINE configs[2]: pretraining_local_point_step = 0.005 (readme.md:109) -> ~7.7e5 lattice points in the two boxes
    n = tr.init_pretraining(batch_size=6144000, lr=0.05, local_point_step=0.005)
    assert 700000 < n < 850000, n
    losses = [float(tr.pretrain_one_epoch()) for _ in range(6)]
    assert losses[-1] < 0.9 * losses[0], losses
    assert torch.equal(student.sigma_net[0].weight, teacher.sigma_net[0].weight)  # MLPs frozen
    # global fine-tuning steps against teacher-rendered targets
    poses = syn.orbit_poses(2, seed=1).cuda()
    g = torch.Generator().manual_seed(0)
    student.mean_count = 0
    for i in range(3):
        r = syn.get_rays(poses[i % 2:i % 2 + 1], syn.lego_intrinsics(), 800, 800, N=4096, generator=g)
        loss = tr.train_step(r["rays_o"].contiguous(), r["rays_d"].contiguous())
        assert torch.isfinite(loss)
    # edited-view render: the teacher shows source content at the target location
    r = syn.get_rays(poses[:1], syn.lego_intrinsics(100, 100), 100, 100)
    img, depth = tr.proxy_truth(r["rays_o"].contiguous(), r["rays_d"].contiguous())
    assert img.shape == (1, 10000, 3) and torch.isfinite(img).all()


def _seal_pair(seed=0):
    from nerf import network
    from sealnerf import SealBBoxMapper, make_student, make_teacher
    torch.manual_seed(seed)
    kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    teacher = make_teacher(network.NeRFNetwork, **kw).cuda()
    student = make_student(network.NeRFNetwork, **kw).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    teacher.density_grid.copy_(torch.from_numpy(grid))
    teacher.density_bitfield.copy_(torch.from_numpy(bits))
    g = torch.Generator().manual_seed(seed)
    for p in teacher.parameters():
        p.data.copy_((torch.rand(p.shape, generator=g) * 0.4 - 0.2))
    student.load_state_dict(teacher.state_dict())
    mapper = SealBBoxMapper(BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    return teacher, student


def test_seal_distillation_graph_replay_matches_eager(hip):
    """configs[2] on the fast path: pretraining epochs replayed from per-chunk HIP graphs give the SAME tables as the eager
    steps (every kernel of the step is deterministic), and the graph-replayed fine-tuning step trains (targets from the
    teacher's proxy render, MSE + L1 depth)."""
    from sealnerf import GraphedSealTrainer, SealTrainer
    res = {}

    def sched(opt):  # main_SealNeRF.py:283-300 steps a LambdaLR after every step: it is what brings the lr back from the
        return torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0)  # pretraining value after `set_lr(-1)`'s cache quirk
    for mode in ("eager", "graph"):
        teacher, student = _seal_pair(0)
        if mode == "eager":
            tr = SealTrainer(student, teacher, lr=1e-2, fp16=True, lr_scheduler=sched)
            tr.graph_pretraining = False
        else:
            tr = GraphedSealTrainer(student, teacher, 4096, lr=1e-2, fp16=True, lr_scheduler=sched)
        n = tr.init_pretraining(batch_size=6144000, lr=0.05, local_point_step=0.005)
        losses = [float(tr.pretrain_one_epoch()) for _ in range(5)]
        res[mode] = (n, losses, student.encoder.embeddings.detach().clone(), student.encoder_color.embeddings.detach().clone(), tr, student)
    assert res["eager"][0] == res["graph"][0]
    assert res["graph"][4]._pt_graphs, "pretraining must have been captured"
    assert res["eager"][1] == res["graph"][1], (res["eager"][1], res["graph"][1])
    assert torch.equal(res["eager"][2], res["graph"][2]) and torch.equal(res["eager"][3], res["graph"][3])
    assert res["graph"][1][-1] < 0.9 * res["graph"][1][0]
    # fine-tuning, graph-replayed
    tr, student = res["graph"][4], res["graph"][5]
    poses = syn.orbit_poses(4, seed=1).cuda()
    g = torch.Generator().manual_seed(0)
    before = student.encoder.embeddings.detach().clone()
    r = syn.get_rays(poses[:1], syn.lego_intrinsics(), 800, 800, N=4096, generator=g)
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    gt = tr.proxy_truth(ro, rd)
    hist = [float(tr.train_step(ro, rd, *gt)) for _ in range(40)]
    assert not torch.equal(before, student.encoder.embeddings)
    # the student starts as a copy of the teacher and has been fitted to the edit: the loss against the teacher's proxy
    # render is small from the first step on and stays there (the depth targets follow the student's depth convention)
    assert tr.n_captures >= 1 and np.isfinite(hist).all() and max(hist) < 2e-2, hist
    # same steps, launched eagerly, on an identically prepared pair
    tr_e, student_e = res["eager"][4], res["eager"][5]
    hist_e = [float(tr_e.train_step(ro, rd, *gt)) for _ in range(40)]
    np.testing.assert_allclose(hist, hist_e, rtol=0.5)  # (different noise streams: same trajectory, not the same numbers)


def test_seal_lr_changes_reach_replayed_graphs_through_the_device_factor(hip):
    """Seal-3D switches learning rates between its phases (`set_lr`, SealNeRF/trainer.py:491-504: pretraining lr, then back).
    Once the fine-tuning step has been captured, lr changes travel through NativeAdam's device-side factor: `set_lr` has to
    leave that factor at new lr / captured lr — a pretraining chunk captured AFTER the fine-tuning graph would otherwise be
    replayed at whatever factor the last phase left behind."""
    from nerf import network
    from sealnerf import GraphedSealTrainer, SealBBoxMapper, make_student, make_teacher
    torch.manual_seed(0)
    kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, log2_hashmap_size=15)
    teacher = make_teacher(network.NeRFNetwork, **kw).cuda()
    student = make_student(network.NeRFNetwork, **kw).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    for m in (teacher, student):
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(bits))
        m.iter_density = 100
    student.load_state_dict(teacher.state_dict())
    mapper = SealBBoxMapper(BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    tr = GraphedSealTrainer(student, teacher, 1024, lr=1e-2, fp16=True, update_extra_interval=16)
    poses = syn.orbit_poses(1, seed=0).cuda()
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=1024, generator=torch.Generator().manual_seed(0))
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    for _ in range(40):                       # 16 eager steps (sample statistics), capture, replays
        tr.train_step(ro, rd)
    opt = tr.optimizer
    assert tr.n_captures >= 1 and opt._lr_captured is not None and float(opt.lr_scale) == 1.0
    tr.init_pretraining(batch_size=6144000, lr=0.05, local_point_step=0.02)
    l0 = float(tr.pretrain_one_epoch())       # eager warm-up step + capture of the chunk's graph, at the pretraining lr
    # (an epoch leaves the pretraining lr set, SealNeRF/trainer.py:363-395; the next fine-tuning step restores it)
    assert abs(float(opt.lr_scale) - 5.0) < 1e-6 and all(g["lr"] == 0.05 for g in opt.param_groups)
    seen = []
    real = type(tr)._pretrain_chunk

    def spy(self, part, k, sl, n_total):
        seen.append(float(opt.lr_scale))      # the factor a replayed chunk graph will read
        return real(self, part, k, sl, n_total)
    type(tr)._pretrain_chunk = spy
    try:
        tr.train_step(ro, rd)                 # a fine-tuning replay in between (closes the pretraining phase: lr back to 1e-2)
        assert float(opt.lr_scale) == 1.0 and all(g["lr"] == 1e-2 for g in opt.param_groups)
        l1 = float(tr.pretrain_one_epoch())   # replays the chunk graph
    finally:
        type(tr)._pretrain_chunk = real
    assert seen and all(abs(f - 0.05 / 1e-2) < 1e-6 for f in seen), seen
    tr.end_pretraining()
    assert float(opt.lr_scale) == 1.0 and np.isfinite([l0, l1]).all()


def test_tensorf_vm48_step_on_gpu(hip):
    from tensoRF import network as trf
    from nerf.trainer import Trainer
    torch.manual_seed(0)
    net = trf.NeRFNetwork(resolution=[128] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    assert net.in_dim == 150 and net.color_net[0].weight.shape == (128, 150)
    nparams = sum(p.numel() for p in net.parameters())
    assert nparams == 3 * 16 * (128 * 128 + 128) + 3 * 48 * (128 * 128 + 128) + 144 * 27 + 150 * 128 + 128 * 128 + 128 * 3
    grid, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(grid))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    net.iter_density = 100
    tr = Trainer(net, lr=2e-2, fp16=True, update_extra_interval=10 ** 9)
    tr.global_step = 1
    poses = syn.orbit_poses(1, seed=0).cuda()
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=4096, generator=torch.Generator().manual_seed(0))
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    gt = torch.rand(4096, 3, device="cuda")
    l0 = float(tr.train_step(ro, rd, gt))
    for _ in range(10):
        l1 = float(tr.train_step(ro, rd, gt))
    assert l1 < l0
    net.upsample_model([160] * 3)
    out = tr.render_image(ro[None, :1000], rd[None, :1000])
    assert out["image"].shape == (1, 1000, 3) and torch.isfinite(out["image"]).all()


def _tensorf_fixture_net():
    import os
    import zlib
    from conftest import GOLDEN
    from tensoRF import network as trf
    T = np.load(os.path.join(GOLDEN, "tensorf.npz"))
    net = trf.NeRFNetwork(resolution=[24, 28, 32], sigma_rank=[4, 5, 6], color_rank=[6, 7, 8], bound=1, cuda_ray=True, density_scale=1,
                          min_near=0.2, density_thresh=10)
    for k, p in net.named_parameters():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) % 1000)
        p.data.copy_(torch.rand(*p.shape, generator=g) - 0.5)
    dens, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(dens))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    return net.cuda(), T


def test_tensorf_train_step_with_l1_term_vs_the_reference_trainer(hip, monkeypatch):
    """tests/golden/tensorf.npz: tensoRF/utils.py `Trainer.train_step` (NGP step + `density_loss() * l1_reg_weight`) EXECUTED
    on the reference's TensoRF network; here the build's trainer on the HIP path in fp32 (fused VM feature kernels, HIP
    marcher / compositing / freq encoder): forward values 1e-5, loss 1e-5, every parameter gradient within 2e-4 of its
    largest element."""
    import raymarching.raymarching as rm
    from tensoRF.utils import Trainer
    net, T = _tensorf_fixture_net()
    with torch.no_grad():
        sg, cl = net(torch.from_numpy(T["fw_x"]).cuda(), torch.from_numpy(T["fw_d"]).cuda())
    np.testing.assert_allclose(sg.cpu().numpy(), T["fw_sigma"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(cl.cpu().numpy(), T["fw_color"], rtol=2e-5, atol=1e-6)
    assert abs(float(net.density_loss()) - float(T["density_loss"])) <= 1e-6 * float(T["density_loss"])
    net.mean_count = 32768
    tr = Trainer(net, lr0=2e-2, lr1=1e-3, l1_reg_weight=float(T["ts_l1_weight"]), fp16=False, update_extra_interval=10 ** 9)
    tr.global_step = 1
    net.train()
    import nerf.renderer as rend
    from test_gpu_golden import _CpuRandom   # the marcher's jitter drawn from the CPU generator, as the fixture's was
    proxy = _CpuRandom()
    monkeypatch.setattr(rm, "torch", proxy)
    monkeypatch.setattr(rend, "torch", proxy)
    torch.manual_seed(5)
    seen = {}
    monkeypatch.setattr(tr, "_reduce_and_step", lambda: seen.update(
        {k: (p.grad if p.grad is not None else getattr(p, "_s3d_grad", None)).detach().float().clone() for k, p in net.named_parameters()}))
    ro, rd, gt = (torch.from_numpy(T[k]).cuda() for k in ("ts_rays_o", "ts_rays_d", "ts_images"))
    loss = tr.train_step(ro[0].contiguous(), rd[0].contiguous(), gt[0].contiguous())
    assert np.array_equal(net.step_counter[0].cpu().numpy(), T["ts_counter"])
    assert abs(float(loss) - float(T["ts_loss"])) <= 1e-5 * float(T["ts_loss"])
    scale = float(tr.scaler.get_scale()) if hasattr(tr.scaler, "get_scale") else 1.0
    for k, g in seen.items():
        key = "ts_grad_" + k.replace(".", "_")
        g = (g / scale).reshape(-1).cpu()
        if key in T.files:
            want = torch.from_numpy(T[key])
        else:
            want, g = torch.from_numpy(T[key + "_at_values"]), g[torch.from_numpy(T[key + "_at"])]
        assert float((g - want).abs().max()) <= 2e-4 * float(want.abs().max()) + 1e-9, k


@pytest.mark.parametrize("graphed", [False, True], ids=["eager", "graph"])
def test_seal_tensorf_teacher_student_pair(hip, graphed):
    """BASELINE configs[4] as main_SealTensoRF.py:14-17 builds it: a TensoRF teacher viewed through the bbox proxy and a
    TensoRF student (`make_teacher` / `make_student`), the student trainer of `get_trainer("tensorf")`: local pretraining
    (L1(sigma) + L1(colour) on the lattice points, nothing frozen on this backbone) and fine-tuning steps against the teacher's
    proxy targets, with the L1 penalty on the density factors inside every step."""
    from sealnerf import SealBBoxMapper, get_trainer, make_student, make_teacher
    from tensoRF import network as trf
    torch.manual_seed(0)
    kw = dict(resolution=[128] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    teacher = make_teacher(trf.NeRFNetwork, **kw).cuda()
    student = make_student(trf.NeRFNetwork, **kw).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    teacher.density_grid.copy_(torch.from_numpy(grid))
    teacher.density_bitfield.copy_(torch.from_numpy(bits))
    teacher.iter_density = 100
    student.load_state_dict(teacher.state_dict())
    student.iter_density = 100
    mapper = SealBBoxMapper(BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    cls = get_trainer("tensorf", graphed=graphed)
    tr = cls(student, teacher, *((4096,) if graphed else ()), lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True, update_extra_interval=10 ** 9)
    assert [g["lr"] for g in tr.optimizer.param_groups] == [2e-2] * 4 + [1e-3] * 2
    n = tr.init_pretraining(batch_size=6144000, lr=0.02, local_point_step=0.01)
    assert 80000 < n < 120000, n
    before = {k: p.detach().clone() for k, p in student.named_parameters()}
    tbefore = {k: p.detach().clone() for k, p in teacher.named_parameters()}
    losses = [float(tr.pretrain_one_epoch()) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0], losses
    assert all(p.requires_grad for p in student.parameters())                 # nothing frozen on the TensoRF backbone
    assert not torch.equal(before["color_net.0.weight"], student.color_net[0].weight)   # ... so its MLP trains too
    # the reference restores group 0's learning rate into EVERY group after pretraining (SealNeRF/trainer.py:491-504)
    assert [g["lr"] for g in tr.optimizer.param_groups] == [2e-2] * 6
    tr.global_step = 1
    poses = syn.orbit_poses(1, seed=0).cuda()
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=4096, generator=torch.Generator().manual_seed(3))
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    gt = tr.proxy_truth(ro, rd)
    assert gt[0].shape == (4096, 3) and torch.isfinite(gt[0]).all() and torch.isfinite(gt[1]).all()
    reg = float(student.density_loss()) * 1e-4
    hist = [float(tr.train_step(ro, rd, *gt)) for _ in range(8)]
    if graphed:  # the sample budget a grid update would set: from here on the student's step is captured and replayed
        student.mean_count = int(student.step_counter[:8, 0].float().mean().item())
        student.local_step = 0
    hist += [float(tr.train_step(ro, rd, *gt)) for _ in range(12 if graphed else 4)]
    if graphed:
        assert tr.graph is not None and tr.n_captures >= 1  # (the student's step was captured and replayed)
    # the student starts as the teacher's copy fitted to the edit: the loss against the proxy targets is small from the first
    # step on and stays there; the penalty is part of it
    assert np.isfinite(hist).all() and max(hist) < 2e-2, hist
    assert min(hist) > 0.5 * reg > 0
    for k, p in teacher.named_parameters():
        assert torch.equal(p, tbefore[k]), k


@pytest.mark.parametrize("N,shape", [(100_000 + 37, (27, 144)), (16 * 1024, (128, 150)), (9000, (3, 128))])
def test_tensorf_tall_linear_matches_nn_linear_under_autocast(hip, N, shape):
    """tensoRF/network.py:_TallLinear (weight gradient as a batched GEMM over 1,024-row chunks, summed in fp32) vs nn.Linear
    under fp16 autocast — the reference's basis_mat / colour-MLP layers (tensoRF/network.py:71-83): same forward values, same
    data gradient, weight gradient within fp16 rounding of a 1e5-term sum."""
    from tensoRF import network as trf
    torch.manual_seed(N)
    layer = torch.nn.Linear(shape[1], shape[0], bias=False).cuda()
    x0 = torch.randn(N, shape[1], device="cuda")
    go = torch.randn(N, shape[0], device="cuda") * 1e-2
    res = []
    for tall in (True, False):
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = trf._linear(layer, x) if tall else layer(x)
            if tall:
                assert y.grad_fn.__class__.__name__.startswith("_TallLinear")
        y.backward(go.to(y.dtype))
        res.append((y.detach().float(), x.grad.clone(), layer.weight.grad.clone()))
    (y_a, gx_a, gw_a), (y_b, gx_b, gw_b) = res
    assert torch.equal(y_a, y_b) and torch.equal(gx_a, gx_b)
    gw_ref = (go.half().double().t() @ x0.half().double()).float()
    scale = float(gw_ref.abs().max())
    assert float((gw_a - gw_ref).abs().max()) <= 2e-3 * scale
    assert float((gw_a - gw_ref).abs().max()) <= float((gw_b - gw_ref).abs().max()) + 1e-3 * scale  # no worse than the library's single GEMM


@pytest.mark.parametrize("kind", ["both", "to", "from"])
def test_seal_bbox_mapper_device_kernel_matches_torch_sequence(hip, kind):
    """csrc/seal.hip vs the torch restatement of SealNeRF/seal_utils.py:132-279 on the same points: identical masks (a point
    exactly on a face could differ by rounding — none may in this random set), mapped points / directions within fp32 rounding."""
    from sealnerf import SealBBoxMapper
    cfg = dict(BBOX, boundType=kind, scale=[1.2, 0.8, 1.0],
               transform=[[0.8, -0.6, 0, 0.3], [0.6, 0.8, 0, 0.05], [0, 0, 1, -0.1], [0, 0, 0, 1]])
    if kind == "both":
        cfg["mapSource"] = [0.9, 0.9, 0.9]
    mapper = SealBBoxMapper(cfg)
    g = torch.Generator().manual_seed(11)
    pts = (torch.rand(200000, 3, generator=g) * 1.6 - 0.8).cuda()
    pts[:7] = 0.0                       # `points.all(1)` rows
    pts[7:20, 1] = 0.0
    dirs = torch.nn.functional.normalize(torch.randn(200000, 3, generator=g), dim=-1).cuda()
    mapper.native = True
    p_n, d_n, m_n = mapper.map_to_origin(pts, dirs)
    mapper.native = False
    p_t, d_t, m_t = mapper.map_to_origin(pts, dirs)
    assert m_n.dtype == torch.bool and 1000 < int(m_t.sum()) < 190000
    assert torch.equal(m_n, m_t)
    torch.testing.assert_close(p_n, p_t, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d_n, d_t, rtol=1e-5, atol=1e-6)
    p_only, none_d, m2 = SealBBoxMapper(cfg).map_to_origin(pts)  # without directions
    assert none_d is None and torch.equal(m2, m_n) and torch.equal(p_only, p_n)


def test_tensorf_vm_features_kernel_matches_grid_sample_sequence(hip):
    """csrc/tensorf.hip vs the reference's op sequence (tensoRF/network.py:112-153: 12 F.grid_sample + stack/cat/mul/sum) on
    the same parameters: non-cubic resolution and unequal ranks (axis / component mix-ups would show), points inside, on the
    border and outside [-1,1] (zeros padding).  fp32, tolerance = a few ulps of the sums (the kernel accumulates the corners
    in the reference kernel's order but without FMA contraction; torch.sum reduces pairwise)."""
    from tensoRF import network as trf
    torch.manual_seed(3)
    net = trf.NeRFNetwork(resolution=[40, 56, 72], sigma_rank=[5, 7, 3], color_rank=[9, 20, 48], bound=1, cuda_ray=True).cuda()
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(50000, 3, generator=g) * 2.6 - 1.3).cuda()
    x[:64] = torch.tensor([-1.0, 1.0, 0.0], device="cuda")      # exactly on the border / centre
    x[64:128] = 1.0
    res = {}
    for fused in (True, False):
        net.fused_vm = fused
        net.zero_grad(set_to_none=True)
        s = net.get_sigma_feat(x)
        c = net.get_color_feat(x)
        (s * torch.linspace(-1, 1, x.shape[0], device="cuda")).sum().backward(retain_graph=True)
        (c * torch.linspace(1, 2, 27, device="cuda")).sum().backward()
        if fused:  # binned backward kernels ran (no autograd graph through grid_sample)
            assert s.grad_fn.name().startswith("_VmFeatures")
        res[fused] = (s.detach(), c.detach(), [p.grad.clone() for p in list(net.sigma_mat) + list(net.sigma_vec) +
                                               list(net.color_mat) + list(net.color_vec) + [net.basis_mat.weight]])
    net.fused_vm = True
    inside = (x.abs() <= 1).all(1)
    assert 0.2 < float(inside.float().mean()) < 0.8
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-5, atol=1e-7)
    assert float(res[False][0][~inside].abs().max()) > 0  # partially outside points still see the in-range corners
    for a, b in zip(res[True][2], res[False][2]):
        # fp32 scatter-adds in a different (and, on both sides, run-dependent) order: the error scales with the sum of the
        # |terms| of an element, not with its (possibly cancelling) value
        torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * float(b.abs().max()))
    # a gradient w.r.t. the coordinates falls back to the grid_sample sequence
    xg = x[:4096].clone().requires_grad_(True)
    net.get_sigma_feat(xg).sum().backward()
    net.fused_vm = False
    xt = x[:4096].clone().requires_grad_(True)
    net.get_sigma_feat(xt).sum().backward()
    net.fused_vm = True
    torch.testing.assert_close(xg.grad, xt.grad, rtol=1e-4, atol=1e-6)
    # under autocast (the -O configs) the features stay fp32 like grid_sample's
    with torch.autocast("cuda", dtype=torch.float16):
        assert net.get_sigma_feat(x).dtype == torch.float32


def test_tensorf_color_features_with_basis_mat_in_the_kernel(hip):
    """s3d_vm_color_forward / _backward (basis_mat applied inside the feature kernels, tensoRF/network.py:149-153) under fp16
    autocast vs the reference's op sequence on the GPU (grid_sample x 12, cat, mul, `.T`, nn.Linear under autocast) and vs the
    two-step native path (product kernel + Linear): outputs to an fp16 ulp of a 144-term sum, gradients of all twelve factors and
    of basis_mat's weight within the tolerance of the unfused kernels; non-cubic resolution, unequal ranks, points outside."""
    from tensoRF import network as trf
    torch.manual_seed(5)
    net = trf.NeRFNetwork(resolution=[40, 56, 72], sigma_rank=[5, 7, 3], color_rank=[9, 20, 48], bound=1, cuda_ray=True).cuda()
    g = torch.Generator().manual_seed(6)
    N = 60000
    x = (torch.rand(N, 3, generator=g) * 2.4 - 1.2).cuda()
    x[:64] = torch.tensor([-1.0, 1.0, 0.0], device="cuda")
    go = (torch.randn(N, 27, generator=g) * 0.05).cuda()
    res = {}
    for mode in ("basis", "kernel+linear", "torch"):
        net.fused_vm, net.fused_basis = mode != "torch", mode == "basis"
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            c = net.get_color_feat(x)
        assert c.dtype == torch.float16 and c.shape == (N, 27)
        if mode == "basis":
            assert c.grad_fn.name().startswith("_VmColorBasis")
        c.backward(go.half())
        res[mode] = (c.detach().float(), [p.grad.float().clone() for p in list(net.color_mat) + list(net.color_vec) + [net.basis_mat.weight]])
    net.fused_vm = net.fused_basis = True
    ref_c, ref_g = res["torch"]
    scale = float(ref_c.abs().max())
    assert scale > 0
    for mode in ("basis", "kernel+linear"):
        c, grads = res[mode]
        assert float((c - ref_c).abs().max()) <= 2e-3 * scale, mode     # fp16 output of a 77-term fp32 sum, fp16 operands
        for a, b in zip(grads, ref_g):
            assert float(b.abs().max()) > 0
            # the product gradient is an fp16 tensor on the reference side (the Linear's data gradient), fp32 inside the kernel
            torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-3 * float(b.abs().max()), msg=lambda m, mode=mode: f"{mode}: {m}")
    # inference: same values without a graph
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert torch.equal(net.get_color_feat(x).float(), res["basis"][0])


def test_tensorf_vm_kernels_match_cpu_oracle(hip):
    """csrc/tensorf.hip (forward + binned backward) vs oracle/vm_features.py (numpy restatement pinned against torch's CPU
    grid_sample in tests/test_vm_oracle.py) on the same inputs"""
    import s3d_hip
    from oracle import vm_features as vo
    g = torch.Generator().manual_seed(9)
    res, ranks, N = [24, 17, 33], [3, 18, 7], 20000
    mat_ids, vec_ids = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)
    planes = [torch.randn(1, ranks[i], res[mat_ids[i][1]], res[mat_ids[i][0]], generator=g) for i in range(3)]
    lines = [torch.randn(1, ranks[i], res[vec_ids[i]], 1, generator=g) for i in range(3)]
    x = torch.rand(N, 3, generator=g) * 2.4 - 1.2
    pn, ln = [p[0].numpy() for p in planes], [l[0, :, :, 0].numpy() for l in lines]
    pd, ld, xd = [p.cuda() for p in planes], [l.cuda() for l in lines], x.cuda()
    rows = sum(ranks)
    out_s = torch.empty(N, device="cuda")
    out_c = torch.empty(rows, N, device="cuda")
    s3d_hip.VmBackend.features_forward(xd, pd, ld, res, True, out_s)
    s3d_hip.VmBackend.features_forward(xd, pd, ld, res, False, out_c)
    np.testing.assert_allclose(out_s.cpu().numpy(), vo.sigma_feat(x.numpy(), pn, ln), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out_c.cpu().numpy(), vo.color_products(x.numpy(), pn, ln), rtol=1e-5, atol=1e-7)
    gs = torch.randn(N, generator=g)
    gc = torch.randn(N, rows, generator=g)
    for reduce, grad, grad_rows in ((True, gs, np.tile(gs.numpy()[None], (rows, 1))), (False, gc, gc.numpy().T)):
        gp, gl = s3d_hip.VmBackend.features_backward(xd, pd, ld, res, reduce, grad.cuda().contiguous())
        rp, rl = vo.factor_grads(x.numpy(), pn, ln, grad_rows)
        for a, b in zip(gp + gl, rp + rl):
            np.testing.assert_allclose(a.cpu().numpy().reshape(b.shape), b, rtol=1e-4, atol=2e-5 * float(np.abs(b).max()))


def test_tensorf_vm_kernels_edge_resolutions(hip):
    """resolution 2 along an axis, points on -1 / 0 / +1 and outside: kernels vs the CPU oracle, forward and backward"""
    import itertools
    import s3d_hip
    from oracle import vm_features as vo
    g = torch.Generator().manual_seed(13)
    mat_ids, vec_ids = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)
    for res in ([2, 5, 3], [7, 2, 2], [4, 4, 9]):
        ranks = [2, 1, 3]
        planes = [torch.randn(1, ranks[i], res[mat_ids[i][1]], res[mat_ids[i][0]], generator=g) for i in range(3)]
        lines = [torch.randn(1, ranks[i], res[vec_ids[i]], 1, generator=g) for i in range(3)]
        corners = torch.tensor(list(itertools.product([-1.0, 0.0, 1.0, -1.0001, 1.0001, 3.0], repeat=3)))
        x = torch.cat([corners, torch.rand(300, 3, generator=g) * 2 - 1]).contiguous()
        N, rows = x.shape[0], sum(ranks)
        pn, ln = [p[0].numpy() for p in planes], [l[0, :, :, 0].numpy() for l in lines]
        pd, ld, xd = [p.cuda() for p in planes], [l.cuda() for l in lines], x.cuda()
        out_s, out_c = torch.empty(N, device="cuda"), torch.empty(rows, N, device="cuda")
        s3d_hip.VmBackend.features_forward(xd, pd, ld, res, True, out_s)
        s3d_hip.VmBackend.features_forward(xd, pd, ld, res, False, out_c)
        np.testing.assert_allclose(out_s.cpu().numpy(), vo.sigma_feat(x.numpy(), pn, ln), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out_c.cpu().numpy(), vo.color_products(x.numpy(), pn, ln), rtol=1e-5, atol=1e-6)
        gc = torch.randn(N, rows, generator=g)
        gp, gl = s3d_hip.VmBackend.features_backward(xd, pd, ld, res, False, gc.cuda().contiguous())
        rp, rl = vo.factor_grads(x.numpy(), pn, ln, gc.numpy().T)
        for a, b in zip(gp + gl, rp + rl):
            np.testing.assert_allclose(a.cpu().numpy().reshape(b.shape), b, rtol=1e-4, atol=2e-5 * float(np.abs(b).max()))


@pytest.mark.parametrize("res", [[300, 300, 300], [24, 17, 33], [2, 5, 3]], ids=["300", "24x17x33", "2x5x3"])
def test_tensorf_backward_bins_native_sort_equals_the_torch_sort_twin(hip, res):
    """s3d_vm_backward_bins (keys + wave-aggregated counts, scan, scatter) against the same list through torch.sort + searchsorted:
    every bin boundary and every bin's set of points (the order inside a bin is free), incl. points without contribution"""
    import s3d_hip
    V = s3d_hip.VmBackend
    g = torch.Generator().manual_seed(4)
    N = 50021
    x = torch.cat([torch.rand(N - 21, 3, generator=g) * 2.3 - 1.15, torch.full((21, 3), 7.0)]).cuda().contiguous()
    planes = [torch.zeros(1, 2, 2, 2, device="cuda") for _ in range(3)]  # (only the ranks are read)
    try:
        V.native_bins = True
        perm, start, nb = V.backward_bins(x, planes, res)
        V.native_bins = False
        perm_t, start_t, nb_t = V.backward_bins(x, planes, res)
    finally:
        V.native_bins = True
    assert nb == nb_t and torch.equal(start, start_t)
    # the same points in every bin: sort each row's (bin of position, point) pairs
    pos = torch.arange(N, device="cuda")
    for r in range(6):
        bin_of_pos = torch.searchsorted(start[r].long().contiguous(), pos, right=True)
        for p in (perm, perm_t):
            assert int(p[r].min()) == 0 and int(p[r].max()) == N - 1 and p[r].unique().numel() == N
        a = torch.sort(bin_of_pos * N + perm[r].long()).values
        b = torch.sort(bin_of_pos * N + perm_t[r].long()).values
        assert torch.equal(a, b), r
    assert int(start[:, -1].min()) <= N - 21 and int(start[:, 0].max()) == 0


def test_tensorf_l1_penalty_inside_the_adam_launch_trains_like_the_autograd_route(hip):
    """tensoRF/utils.py `l1_in_update`: the L1 term as a value + its gradient formed inside NativeAdam's launch, against the term
    through autograd (sign / scale / accumulate passes) — same steps, same RNG: the losses agree and after three steps all but a
    vanishing share of the density factors' elements are equal to rounding (an element whose total gradient is ~0 may take its
    Adam-normalised step in the other direction)"""
    from tensoRF import network as trf
    from tensoRF.utils import Trainer
    from nerf import synthetic as syn
    poses = syn.orbit_poses(2, seed=0).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    res = {}
    for in_update in (True, False):
        torch.manual_seed(3)
        net = trf.NeRFNetwork(resolution=[64] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
        net.density_grid.copy_(torch.from_numpy(grid)); net.density_bitfield.copy_(torch.from_numpy(bits))
        net.iter_density = 100
        tr = Trainer(net, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-2, fp16=True, update_extra_interval=10 ** 9, native_optim=True)
        tr.l1_in_update = in_update
        tr.global_step = 1
        losses = []
        for k in range(3):
            r = syn.get_rays(poses[k % 2:k % 2 + 1], syn.lego_intrinsics(), 800, 800, N=1024, generator=torch.Generator().manual_seed(k))
            torch.manual_seed(100 + k)
            losses.append(float(tr.train_step(r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous(),
                                              torch.rand(1024, 3, generator=torch.Generator().manual_seed(50 + k)).cuda())))
        res[in_update] = (losses, [p.detach().clone() for p in list(net.sigma_mat) + list(net.sigma_vec)])
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 1e-4 * abs(b), (res[True][0], res[False][0])
    moved = False
    for a, b in zip(res[True][1], res[False][1]):
        off = ((a - b).abs() > 1e-5 + 1e-4 * b.abs()).float().mean()
        assert float(off) < 2e-3, float(off)
        moved |= bool((a != 0).any())
    assert moved


def test_tensorf_colour_mlp_on_the_ffmlp_kernels_matches_the_linear_chain(hip):
    """tensoRF/network.py `fused_mlp`: the colour MLP 150 -> 128 -> 128 -> 3 (input padded to 160) through the W = 128 MFMA kernels of
    the ffmlp package against the nn.Linear chain under fp16 autocast — colours, and the gradients of every MLP weight, of
    basis_mat and of the colour factors (the data gradient travels on through the encoders and the VM kernels)"""
    from tensoRF import network as trf
    torch.manual_seed(1)
    net = trf.NeRFNetwork(resolution=[48] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    with torch.no_grad():
        for l in net.color_net:
            l.weight.mul_(3.0)  # (default init leaves the colours near 0.5: make the network matter)
    g = torch.Generator().manual_seed(2)
    N = 128 * 40
    x = (torch.rand(N, 3, generator=g) * 1.8 - 0.9).cuda()
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
    go = torch.randn(N, 3, generator=g).cuda()
    out = {}
    for fused in (True, False):
        net.fused_mlp = fused
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb = net(x, d)
        assert rgb.shape == (N, 3)
        (rgb.float() * go).sum().backward()
        out[fused] = (rgb.detach().float(), {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None})
    net.fused_mlp = True
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        _, rgb_eval = net(x, d)  # (rendering: the MLP's inference entry point, no activation buffers)
    assert torch.equal(rgb_eval.float(), out[True][0])
    torch.testing.assert_close(out[True][0], out[False][0], rtol=0, atol=2e-3)
    assert float((out[True][0] - 0.5).abs().max()) > 0.2
    keys = set(out[True][1]) & set(out[False][1])
    assert any(k.startswith("color_net.0") for k in keys) and any(k.startswith("color_mat") for k in keys) and "basis_mat.weight" in keys
    for k in keys:
        if k.startswith("sigma"):
            continue
        a, b = out[True][1][k], out[False][1][k]
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-6, (k, float((a - b).abs().max()), float(b.abs().max()))


def test_tensorf_factor_backward_raises_the_scalers_flag_itself(hip):
    """tensoRF/utils.py `_attach_source_checks`: with the native GradScaler on one replica the factor backward marks the gradients
    it wrote and raises found_inf for a non-finite bound, so the scaler's check pass reads only the MLP weights; a non-finite
    factor then skips the whole step (no parameter moves, the scale backs off) exactly like the scaler's own pass would"""
    from tensoRF import network as trf
    from tensoRF.utils import Trainer
    from nerf import synthetic as syn
    import nerf.optim as optim
    poses = syn.orbit_poses(1, seed=0).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    torch.manual_seed(3)
    net = trf.NeRFNetwork(resolution=[64] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    net.density_grid.copy_(torch.from_numpy(grid)); net.density_bitfield.copy_(torch.from_numpy(bits))
    net.iter_density = 100
    tr = Trainer(net, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True, update_extra_interval=10 ** 9, native_optim=True)
    assert net._s3d_found_inf is tr.scaler._found_inf
    tr.global_step = 1
    r = syn.get_rays(poses[:1], syn.lego_intrinsics(), 800, 800, N=1024, generator=torch.Generator().manual_seed(0))
    batch = (r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous(), torch.rand(1024, 3).cuda())
    seen = []
    inner = torch._amp_foreach_non_finite_check_and_unscale_

    def spy(grads, found_inf, inv_scale):
        seen.append(sum(g.numel() for g in grads))
        return inner(grads, found_inf, inv_scale)
    torch._amp_foreach_non_finite_check_and_unscale_ = spy
    try:
        tr.train_step(*batch)
        factors = sum(p.numel() for p in list(net.sigma_mat) + list(net.sigma_vec) + list(net.color_mat) + list(net.color_vec))
        assert seen and max(seen) < factors, (seen, factors)  # the pass no longer covers the factors
        assert all("_s3d_grad_checked" not in p.__dict__ for p in net.parameters())
        # a non-finite colour line factor: found_inf from the kernels, the step is skipped
        before = [p.detach().clone() for p in net.parameters()]
        scale0 = float(tr.scaler.get_scale())
        with torch.no_grad():
            net.color_vec[1][0, 3, 7, 0] = float("inf")
            before[[id(p) for p in net.parameters()].index(id(net.color_vec[1]))][0, 3, 7, 0] = float("inf")
        tr.train_step(*batch)
        for a, b in zip(net.parameters(), before):
            assert torch.equal(a.detach(), b)
        assert float(tr.scaler.get_scale()) < scale0
    finally:
        torch._amp_foreach_non_finite_check_and_unscale_ = inner


def test_tensorf_small_fused_launches_match_their_torch_expressions(hip):
    """s3d_aabb_normalize == 2 (x - lo) / (hi - lo) - 1 bit for bit (same operations, same order); s3d_weighted_abs_sum == the sum of
    mean|t| over the density factors to fp32 rounding; the colour head through _NgpRgb == torch.sigmoid on the fp16 output"""
    import s3d_hip
    from tensoRF import network as trf
    g = torch.Generator().manual_seed(8)
    x = (torch.rand(5000, 3, generator=g) * 3 - 1.5).cuda()
    aabb = torch.tensor([-1.0, -0.7, -1.3, 0.9, 1.1, 0.8], device="cuda")
    out = torch.empty_like(x)
    s3d_hip.VmBackend.aabb_normalize(x, aabb, out)
    assert torch.equal(out, 2 * (x - aabb[:3]) / (aabb[3:] - aabb[:3]) - 1)
    torch.manual_seed(2)
    net = trf.NeRFNetwork(resolution=[40, 33, 28], bound=1, cuda_ray=True).cuda()
    a = float(net.density_loss_value())
    net.fused_l1 = False
    b = float(net.density_loss())
    assert abs(a - b) <= 2e-6 * abs(b), (a, b)
    h = torch.randn(4096, 16, generator=g).half().cuda()
    from nerf.network_ff import _NgpRgb
    assert torch.equal(_NgpRgb.apply(h), torch.sigmoid(h[:, :3]).float())


# ------------------------------------------------------------------------------------------------ configs[4] at its real shape
def _vm48_marched_samples(n_rays=6144, seed=0):
    """ray-ordered marched samples of the synthetic scene (what the training step hands the VM kernels): > 1e5 points"""
    import raymarching
    grid, bits = syn.lego_like_density_grid(seed=0)
    poses = syn.orbit_poses(2, seed=seed).cuda()
    r = syn.get_rays(poses[:1], syn.lego_intrinsics(), 800, 800, N=n_rays, generator=torch.Generator().manual_seed(seed))
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device="cuda")
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, torch.from_numpy(bits).cuda(), 1, 128, nears, fars, counter, 0,
                                                            False, 128, True, 0, 1024)
    m = int(counter[0])
    return xyzs[:m].contiguous(), m


def _vm48(res=300, seed=3):
    from tensoRF import network as trf
    torch.manual_seed(seed)
    net = trf.NeRFNetwork(resolution=[res] * 3, bound=1, cuda_ray=True).cuda()   # the defaults ARE VM-48: sigma 16x3, colour 48x3
    assert [p.shape[1] for p in net.sigma_mat] == [16] * 3 and [p.shape[1] for p in net.color_mat] == [48] * 3
    return net


def _factor_grads(net):
    return [p.grad.float().clone() for p in list(net.sigma_mat) + list(net.sigma_vec) + list(net.color_mat) + list(net.color_vec) + [net.basis_mat.weight]]


def _abs_sum_bound(net, x, g_abs_s, g_abs_c):
    """per-element sum of |terms| of each factor gradient: the torch sequence run on |parameters| with |upstream gradients| —
    the scale rounding errors of ANY summation order are proportional to"""
    import copy
    a = copy.deepcopy(net)
    a.fused_vm = False
    with torch.no_grad():
        for p in a.parameters():
            p.abs_()
    a.zero_grad(set_to_none=True)
    (a.get_sigma_feat(x) * g_abs_s).sum().backward()
    cf = a.get_color_feat(x)
    (cf * g_abs_c).sum().backward()
    return _factor_grads(a)


@pytest.mark.parametrize("case", ["marched", "hot_tile"])
def test_vm48_resolution300_backward_vs_grid_sample_autograd(hip, case):
    """configs[4]'s real shape — sigma rank 16x3, colour rank 48x3 + basis_mat, resolution 300, > 1e5 ray-ordered marched
    samples — s3d_vm_features_backward / s3d_vm_color_backward against F.grid_sample's autograd in fp32 on the same parameters
    (tensoRF/network.py:112-153).  `hot_tile`: every point inside ONE 8x8 plane tile (a cell takes ~1e5 / 64 contributions per
    range: the fixed-point accumulators' worst case).
    Tolerance: both sides sum ~1e3 - 1e5 fp32 terms per cell in different orders; the error of either is bounded by
    (terms) x 2^-24 x sum|terms|, measured here as 2e-6 x the per-element sum of |terms| (the fixed-point path itself rounds each
    term at 2^-50 of the batch bound: far below)."""
    net = _vm48()
    if case == "marched":
        x, m = _vm48_marched_samples()
        assert m >= 100000, m
    else:
        g = torch.Generator().manual_seed(5)
        cell = 2.0 / 299
        x = (torch.rand(120000, 3, generator=g) * (6 * cell) + (-1.0 + (12 * 8 + 1) * cell)).cuda()  # cells 97..103 on every axis: tile (12, 12)
        m = x.shape[0]
    g = torch.Generator().manual_seed(7)
    gs = torch.randn(m, generator=g).cuda()
    gc = (torch.randn(m, 27, generator=g) * 0.1).cuda()
    res = {}
    for fused in (True, False):
        net.fused_vm = net.fused_basis = fused
        net.zero_grad(set_to_none=True)
        s = net.get_sigma_feat(x)
        (s * gs).sum().backward()
        c = net.get_color_feat(x)
        if fused:
            assert s.grad_fn.name().startswith("_VmFeatures")
        (c.float() * gc).sum().backward()
        res[fused] = (s.detach(), c.detach().float(), _factor_grads(net))
    net.fused_vm = net.fused_basis = True
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-4, atol=1e-5)
    bound = _abs_sum_bound(net, x, gs.abs(), gc.abs())
    names = [f"sigma_mat{i}" for i in range(3)] + [f"sigma_vec{i}" for i in range(3)] + [f"color_mat{i}" for i in range(3)] + \
            [f"color_vec{i}" for i in range(3)] + ["basis_mat"]
    for name, a, b, bd in zip(names, res[True][2], res[False][2], bound):
        assert float(b.abs().max()) > 0, name
        err = (a - b).abs()
        tol = 2e-6 * bd + 1e-6 * float(b.abs().max())
        assert bool((err <= tol).all()), (name, float((err / tol).max()), float(err.max()), float(b.abs().max()))
        assert bool(((a != 0) == (b != 0)).all() or (err[(a != 0) != (b != 0)] <= tol[(a != 0) != (b != 0)]).all()), name


@pytest.mark.parametrize("sigma_rank,color_rank,res,fused_basis", [(32, 64, 96, True), (48, 32, 72, False), (64, 48, 72, True)])
def test_vm_matrix_core_backward_other_ranks_vs_grid_sample_autograd(hip, sigma_rank, color_rank, res, fused_basis):
    """The 32-points-per-trip factor backward (k_vm_plane_backward_mm / k_vm_line_backward_mm: ranks 32 / 48 / 64) on the rank
    blocks VM-48 does not use and in its three modes: density features summed over the ranks (MODE 0, sigma rank >= 32), colour
    products handed back un-reduced (MODE 1: fused_basis off), colour behind basis_mat on the matrix cores (MODE 2) — against
    F.grid_sample's autograd on ray-ordered marched samples, tolerance as in the VM-48 test above."""
    from tensoRF import network as trf
    torch.manual_seed(11)
    net = trf.NeRFNetwork(resolution=[res] * 3, sigma_rank=[sigma_rank] * 3, color_rank=[color_rank] * 3, bound=1, cuda_ray=True).cuda()
    x, m = _vm48_marched_samples(n_rays=2048)
    g = torch.Generator().manual_seed(13)
    gs = torch.randn(m, generator=g).cuda()
    gc = (torch.randn(m, 27, generator=g) * 0.1).cuda()
    res_ = {}
    for fused in (True, False):
        net.fused_vm = fused
        net.fused_basis = fused and fused_basis
        net.zero_grad(set_to_none=True)
        (net.get_sigma_feat(x) * gs).sum().backward()
        (net.get_color_feat(x).float() * gc).sum().backward()
        res_[fused] = _factor_grads(net)
    net.fused_vm = net.fused_basis = True
    bound = _abs_sum_bound(net, x, gs.abs(), gc.abs())
    for k, (a, b, bd) in enumerate(zip(res_[True], res_[False], bound)):
        assert float(b.abs().max()) > 0, k
        err = (a - b).abs()
        tol = 2e-6 * bd + 1e-6 * float(b.abs().max())
        if k == 12 and not fused_basis:
            tol = tol + 2e-3 * float(b.abs().max())  # (basis_mat through the fp16 nn.Linear's own autograd on both sides)
        assert bool((err <= tol).all()), (k, float((err / tol).max()), float(err.max()), float(b.abs().max()))


def test_tensorf_padded_sample_batch_follows_the_device_side_count(hip):
    """s3d_hip.row_limit -> n_valid on the TensoRF sample path (VM feature kernels, binning, factor backward, frequency packing,
    W = 128 MLP, rgb head): a batch padded to a static extent with NaN positions behind the announced count gives the outputs of
    the live rows bit for bit and the same parameter gradients as the unpadded call (factor gradients: exact fixed-point sums,
    fp32 flush order; MLP weights: the partial-sum split follows the batch extent, 2e-3)."""
    import s3d_hip
    net = _vm48(res=96)
    x, m = _vm48_marched_samples(n_rays=1024)
    n0 = (m // 128) * 128 - 300 - 37            # the count the marcher would leave on the device
    live = (n0 + 127) // 128 * 128              # rows the kernels work on
    M = live + 2048                             # the static extent of the padded buffers
    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1).cuda()
    ws, wr = torch.randn(live, generator=g).cuda(), torch.randn(live, 3, generator=g).cuda()
    xp = torch.full((M, 3), float("nan"), device="cuda")
    xp[:live] = x[:live]
    counter = torch.tensor([n0, 0], dtype=torch.int32, device="cuda")
    res = {}
    for tag in ("plain", "padded"):
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            if tag == "plain":
                sigma, rgb = net(x[:live].contiguous(), d[:live].contiguous())
            else:
                with s3d_hip.row_limit(counter, M):
                    sigma, rgb = net(xp, d)
        ((sigma[:live].float() * ws).sum() + (rgb[:live].float() * wr).sum()).backward()
        res[tag] = (sigma[:live].detach().float().clone(), rgb[:live].detach().float().clone(),
                    {n: p.grad.float().clone() for n, p in net.named_parameters() if p.grad is not None})
    assert torch.equal(res["plain"][0], res["padded"][0]) and torch.equal(res["plain"][1], res["padded"][1])
    assert set(res["plain"][2]) == set(res["padded"][2]) and len(res["plain"][2]) >= 13
    for n, a in res["plain"][2].items():
        b = res["padded"][2][n]
        assert bool(torch.isfinite(b).all()), n
        tol = (2e-3 if n.startswith("color_net") else 1e-5) * float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= tol, (n, float((a - b).abs().max()), float(a.abs().max()))


def test_tensorf_rank_fastest_shadows_give_the_same_bits(hip):
    """s3d_vm_transpose_factors + planes_t / lines_t: the feature kernels read a corner's rank channels as one contiguous run
    ([H][W][R] / [Dn][R] shadows taken from the parameters at every forward) — the same values through the same arithmetic, so
    density features and colour features equal the run on the parameters' own [R][H][W] layout bit for bit, the factor gradients
    within the fp32 order of their few atomics."""
    import s3d_hip
    net = _vm48(res=96)
    x, m = _vm48_marched_samples(n_rays=768)
    mats, vecs = [p.detach().contiguous() for p in net.color_mat], [p.detach().contiguous() for p in net.color_vec]
    pt, lt = s3d_hip.VmBackend.transpose_factors(mats, vecs, net.resolution)
    for a, t in zip(mats, pt):
        assert torch.equal(t, a[0].permute(1, 2, 0).contiguous())
    for a, t in zip(vecs, lt):
        assert torch.equal(t, a[0, :, :, 0].t().contiguous())
    g = torch.Generator().manual_seed(21)
    gs, gc = torch.randn(m, generator=g).cuda(), (torch.randn(m, 27, generator=g) * 0.1).cuda()
    res = {}
    for on in (True, False):
        net.fused_shadows = on
        net.zero_grad(set_to_none=True)
        s = net.get_sigma_feat(x)
        (s * gs).sum().backward()
        c = net.get_color_feat(x)
        (c.float() * gc).sum().backward()
        res[on] = (s.detach().clone(), c.detach().clone(), _factor_grads(net))
    net.fused_shadows = True
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    # (the gradients: the same exact tile sums, but tiles split over workgroups and basis_mat's gradient leave through fp32 atomics,
    #  whose order differs from run to run)
    for a, b in zip(res[True][2], res[False][2]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12


def test_vm_backward_keeps_gradients_far_below_the_batch_maximum(hip):
    """the fixed-point accumulators are scaled by ONE bound per call (placed at 2^50, contributions rounded to
    nearest).  A region whose gradients lie 2^36 below the batch maximum — and, with the bound's slack (max |g| x max |line|),
    more than 2^40 below the bound — still receives every one of them: compared with the grid_sample autograd on the same
    inputs, cell by cell, relative to each cell's own sum of |terms|."""
    net = _vm48(res=64)
    g = torch.Generator().manual_seed(11)
    m = 40000
    x = (torch.rand(m, 3, generator=g) * 1.9 - 0.95).cuda()
    gs = torch.randn(m, generator=g).cuda()
    tiny = x[:, 0] < 0  # the left half of space: gradients 2^-36 of the right half's
    gs = torch.where(tiny, gs * 2.0 ** -36, gs)
    res = {}
    for fused in (True, False):
        net.fused_vm = fused
        net.zero_grad(set_to_none=True)
        (net.get_sigma_feat(x) * gs).sum().backward()
        res[fused] = [p.grad.clone() for p in list(net.sigma_mat) + list(net.sigma_vec)]
    net.fused_vm = True
    # plane 0 spans (x, y): its columns at x < 0 see only the tiny gradients
    a, b = res[True][0][0], res[False][0][0]            # [16, H(y), W(x)]
    left = slice(0, 28)
    assert float(b[:, :, left].abs().max()) < 2.0 ** -20 * float(b.abs().max()) and float(b[:, :, left].abs().max()) > 0
    nz = b[:, :, left] != 0
    # (a contribution under half a quantum — 2^-51 of max |g| x max |line| — rounds to zero: cells fed by such terms alone
    #  may come out empty; they are a fraction of a percent here and their sums are below the absolute floor used below)
    assert float(((a[:, :, left] != 0) & nz).sum()) >= 0.99 * float(nz.sum()), "cells fed only by small gradients must still receive them"
    line_max = max(float(p.abs().max()) for p in net.sigma_vec)
    floor = 2.0 ** -50 * float(gs.abs().max()) * line_max * 64  # 64 contributions' worth of rounding at the quantum
    # error relative to each cell's OWN sum of |terms| (a cell's value may cancel; its terms do not): the torch sequence on
    # |parameters| with |gradients|
    import copy
    ab = copy.deepcopy(net)
    ab.fused_vm = False
    with torch.no_grad():
        for p in ab.parameters():
            p.abs_()
    ab.zero_grad(set_to_none=True)
    (ab.get_sigma_feat(x) * gs.abs()).sum().backward()
    bd = ab.sigma_mat[0].grad[0][:, :, left]
    err = (a[:, :, left] - b[:, :, left]).abs()
    assert bool((err <= 2e-3 * bd + floor).all()), float((err / (2e-3 * bd + floor)).max())  # (>= 9 bits per contribution at this depth)
    for a, b in zip(res[True], res[False]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * float(b.abs().max()))


def test_vm_backward_non_finite_inputs_raise_the_scaler_flag(hip):
    """a non-finite upstream gradient / line factor: the bound words catch it where the gradient is written — `found_inf` is
    raised (GradScaler skips the step, like the reference's unscale pass over the .grad tensors) and nothing hangs"""
    import s3d_hip
    net = _vm48(res=48)
    g = torch.Generator().manual_seed(13)
    m = 20000
    x = (torch.rand(m, 3, generator=g) * 2 - 1).cuda()
    for poison in ("grad", "line"):
        flag = torch.zeros(1, device="cuda")
        net._s3d_found_inf = flag
        net.zero_grad(set_to_none=True)
        gs = torch.randn(m, generator=g).cuda()
        saved = net.sigma_vec[1].data[0, 3, 5, 0].clone()
        if poison == "grad":
            gs[777] = float("inf")
        else:
            net.sigma_vec[1].data[0, 3, 5, 0] = float("nan")
        (net.get_sigma_feat(x) * gs).sum().backward()
        torch.cuda.synchronize()
        net.sigma_vec[1].data[0, 3, 5, 0] = saved
        assert float(flag) == 1.0, poison
    net._s3d_found_inf = None
    flag = torch.zeros(1, device="cuda")
    net._s3d_found_inf = flag
    net.zero_grad(set_to_none=True)
    (net.get_sigma_feat(x) * torch.randn(m, generator=g).cuda()).sum().backward()
    assert float(flag) == 0.0 and all(torch.isfinite(p.grad).all() for p in list(net.sigma_mat) + list(net.sigma_vec))
