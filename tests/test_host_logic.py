"""CPU: host logic of the build's wrappers / renderer / network / data-parallel layer, driven by the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO


def test_grid_encoder_lego_layout(oracle_wrappers):
    """offset table and parameter count of the Lego config (grid.py:117-131)"""
    enc = oracle_wrappers.gg.GridEncoder(desired_resolution=2048)
    assert enc.offsets.tolist()[:7] == [0, 4920, 18744, 51512, 136696, 352696, 876984]
    assert int(enc.offsets[-1]) == 6119864 and enc.embeddings.shape == (6119864, 2)
    assert abs(enc.per_level_scale - 1.381912879967776) < 1e-12
    assert enc.output_dim == 32 and float(enc.embeddings.abs().max()) <= 1e-4


def test_march_wrapper_contracts(oracle_wrappers):
    """padding rule `m += align - m % align`, zero tails, budgeted M, counter semantics (raymarching.py:161-235)"""
    from nerf import synthetic as syn
    rm = oracle_wrappers.rm
    _, bits = syn.lego_like_density_grid(seed=0)
    bits = torch.from_numpy(bits)
    poses = syn.orbit_poses(1, seed=0)
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800, N=1024, generator=torch.Generator().manual_seed(0))
    ro, rd = r["rays_o"][0], r["rays_d"][0]
    nears, fars = rm.near_far_from_aabb(ro, rd, torch.tensor([-1.0, -1, -1, 1, 1, 1]), 0.2)
    counter = torch.zeros(2, dtype=torch.int32)
    xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, counter, -1, False, 128, False, 0, 1024)
    m = int(counter[0])
    assert xyzs.shape[0] == m + 128 - m % 128 and (xyzs[m:] == 0).all() and int(counter[1]) == 1024
    # budgeted call: M = aligned mean_count, overflowing rays dropped but still counted
    counter2 = torch.zeros(2, dtype=torch.int32)
    budget = m // 2
    x2, _, _, rays2 = rm.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, counter2, budget, False, 128, False, 0, 1024)
    assert x2.shape[0] == budget + 128 - budget % 128 and int(counter2[0]) == m
    assert torch.equal(rays2, rays)
    fits = (rays2[:, 1] + rays2[:, 2]) <= x2.shape[0]
    k = int(torch.nonzero(~fits)[0])
    assert (x2[int(rays2[k - 1, 1] + rays2[k - 1, 2]):] == 0).all()
    # composite through autograd
    sig = torch.rand(xyzs.shape[0], requires_grad=True)
    rgb = torch.rand(xyzs.shape[0], 3, requires_grad=True)
    ws, depth, img = rm.composite_rays_train(sig, rgb, deltas, rays, 1e-4)
    (img.sum() + ws.sum()).backward()
    assert sig.grad.shape == sig.shape and (sig.grad[m:] == 0).all()


def _tiny_net(oracle_wrappers, ff=False):
    from nerf import network, network_ff
    torch.manual_seed(0)
    mod = network_ff if ff else network
    net = mod.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    return net


def test_network_forward_and_param_count(oracle_wrappers):
    net = _tiny_net(oracle_wrappers)
    n = sum(p.numel() for p in net.parameters())
    assert n == 24490848  # 2 x 12,239,728 + 3,072 + 8,320
    keys = set(net.state_dict().keys())
    for k in ("encoder.embeddings", "encoder.offsets", "encoder_color.embeddings", "sigma_net.0.weight", "sigma_net.1.weight",
              "color_net.0.weight", "color_net.2.weight", "density_grid", "density_bitfield", "step_counter"):
        assert k in keys
    x = torch.rand(300, 3) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(300, 3), dim=-1)
    sigma, rgb = net(x, d)
    assert sigma.shape == (300,) and rgb.shape == (300, 3) and (sigma > 0).all() and ((rgb > 0) & (rgb < 1)).all()
    assert torch.equal(net.density(x)["sigma"], sigma)


def test_renderer_train_eval_and_density_update(oracle_wrappers):
    """run_cuda training branch, inference loop (host compaction) and update_extra_state on the oracle backend"""
    from nerf import synthetic as syn
    net = _tiny_net(oracle_wrappers)
    # fake a converged occupancy grid from the synthetic scene
    grid, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(grid))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    poses = syn.orbit_poses(1, seed=0)
    r = syn.get_rays(poses, syn.lego_intrinsics(64, 64), 64, 64)
    ro, rd = r["rays_o"], r["rays_d"]
    net.train()
    out = net.render(ro, rd, bg_color=1, perturb=True, max_steps=1024)
    assert out["image"].shape == (1, 4096, 3) and int(net.step_counter[0, 0]) > 0 and net.local_step == 1
    out["image"].sum().backward()
    assert net.encoder.embeddings.grad is not None and net.encoder_color.embeddings.grad.abs().sum() > 0
    net.eval()
    with torch.no_grad():
        ev = net.render(ro, rd, bg_color=1, perturb=False, max_steps=1024)
    assert ev["image"].shape == (1, 4096, 3) and torch.isfinite(ev["image"]).all()
    # untouched background rays composite to the background colour
    miss = out["weights_sum"].detach() == 0
    assert miss.any() and torch.allclose(ev["image"][0][miss], torch.ones(3))
    # density update: first call = full sweep over 128^3 cells
    net.train()
    torch.manual_seed(0)
    net.reset_extra_state()
    net.local_step = 3
    net.step_counter[:3, 0] = torch.tensor([100, 200, 300], dtype=torch.int32)
    net.update_extra_state()
    assert net.iter_density == 1 and net.mean_count == 200 and net.local_step == 0
    assert net.density_bitfield.shape == (128 ** 3 // 8,) and net.mean_density > 0
    ref_bits = np.packbits((net.density_grid.numpy().reshape(-1) > np.float32(min(net.mean_density, net.density_thresh))).astype(np.uint8),
                           bitorder="little")
    assert np.array_equal(ref_bits, net.density_bitfield.numpy())


def test_ffmlp_module_padding(oracle_wrappers):
    ff = oracle_wrappers.ff
    net = ff.FFMLP(32, 3, 64, 3)
    assert net.num_parameters == 64 * (32 + 64 * 2 + 16) and net.padded_output_dim == 16
    torch.manual_seed(42)
    ref = torch.empty(net.num_parameters).uniform_(-np.sqrt(3 / 64), np.sqrt(3 / 64))
    assert torch.equal(net.weights.data, ref)
    x = torch.randn(200, 32)
    net.train()
    y = net(x.half())
    assert y.shape == (200, 3)
    net.eval()
    assert torch.equal(net(x.half()), y)


def test_shard_slice_partition():
    from parallel import shard_slice
    for n in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            parts = [shard_slice(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts[:-1], parts[1:]))
            assert max(p[1] - p[0] for p in parts) - min(p[1] - p[0] for p in parts) <= 1


_DP_SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["S3D_REPO"]); sys.path.insert(0, os.path.join(os.environ["S3D_REPO"], "seal-3d_amd"))
from parallel import RayShardedDP, init_from_env, shard_slice
rank, world, _ = init_from_env("gloo")
torch.manual_seed(100 + rank)            # replicas start DIFFERENT; register() must broadcast rank 0's
model = torch.nn.Sequential(torch.nn.Linear(8, 16, bias=False), torch.nn.ReLU(), torch.nn.Linear(16, 3, bias=False))
dp = RayShardedDP().register(model)
g = torch.Generator().manual_seed(7)
x, y = torch.randn(64, 8, generator=g), torch.randn(64, 3, generator=g)
lo, hi = shard_slice(64, rank, world)
loss = torch.nn.functional.mse_loss(model(x[lo:hi]), y[lo:hi], reduction="sum") / (64 * 3) * world
loss.backward()
dp.allreduce_grads()
flat = dp.flat.clone()
# single-process reference on the full batch with rank 0's weights
ref = torch.nn.Sequential(torch.nn.Linear(8, 16, bias=False), torch.nn.ReLU(), torch.nn.Linear(16, 3, bias=False))
ref.load_state_dict(model.state_dict())
torch.nn.functional.mse_loss(ref(x), y).backward()
ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
ok = torch.allclose(flat, ref_flat, rtol=1e-5, atol=1e-6)
w = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(w, torch.cat([p.data.reshape(-1) for p in model.parameters()]))
same = all(torch.equal(w[0], t) for t in w)
views = all(p.grad.data_ptr() >= dp.flat.data_ptr() for p in model.parameters())
print(f"RANK{rank} ok={ok} same={same} views={views}")
dist.destroy_process_group()
'''


def test_ray_sharded_dp_gloo_world2(tmp_path):
    """world_size-2 gloo run: sharded-batch gradient after the flat-bucket all-reduce == full-batch gradient"""
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT)
    env = dict(os.environ, S3D_REPO=REPO, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    for r in (0, 1):
        assert f"RANK{r} ok=True same=True views=True" in res.stdout, res.stdout + res.stderr[-1500:]


def test_sync_free_partial_grid_update_matches_reference_rule(oracle_wrappers):
    """partial_grid_update_device / finish_extra_state (the capturable steady-state occupancy update) against
    update_extra_state's rule on an analytic density: occupied picks are occupied and uniform, the EMA-max update and
    the bitfield agree with the reference sequence up to the jitter of the random samples."""
    import torch
    from nerf import renderer, synthetic as syn
    lo, hi = syn.lego_like_boxes(0)

    class Analytic(renderer.NeRFRenderer):
        def density(self, x):
            return {"sigma": syn.box_density(x, lo, hi, sigma=40.0)}

    def make():
        R = Analytic(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
        R.grid_size = 32  # small grid: CPU test
        R.density_grid = torch.zeros(1, 32 ** 3)
        R.density_bitfield = torch.zeros(32 ** 3 // 8, dtype=torch.uint8)
        return R
    torch.manual_seed(0)
    A, B = make(), make()
    grid0 = torch.rand(1, 32 ** 3)
    grid0[grid0 < 0.7] = 0          # 30 % occupied
    grid0[0, :100] = -1             # never-seen cells stay untouched
    A.density_grid.copy_(grid0)
    B.density_grid.copy_(grid0)
    occ = B._pick_occupied(0, 200000)
    assert bool((grid0[0, occ] > 0).all())
    n_occ = int((grid0[0] > 0).sum())
    hist = torch.bincount(occ, minlength=32 ** 3)[grid0[0] > 0].float()
    assert hist.min() > 0 and abs(float(hist.mean()) - 200000 / n_occ) < 1e-3 and float(hist.std()) < 2.0 * (200000 / n_occ) ** 0.5
    # steady state: start both from a full sweep of the analytic field (jitter only moves cells on the box faces)
    A.density_grid.zero_()
    A.iter_density = 0
    A.update_extra_state()
    A.density_grid[0, :100] = -1
    B.density_grid.copy_(A.density_grid)
    A.iter_density = B.iter_density = 16
    A.local_step = B.local_step = 3
    A.step_counter[:3, 0] = torch.tensor([100, 200, 303], dtype=torch.int32)
    B.step_counter.copy_(A.step_counter)
    for _ in range(6):
        A.local_step = B.local_step = 3
        A.update_extra_state()
        B.finish_extra_state(B.partial_grid_update_device())
    assert A.iter_density == B.iter_density == 22 and A.mean_count == B.mean_count == 201 and B.local_step == 0
    assert bool((B.density_grid[0, :100] == -1).all())
    assert abs(A.mean_density - B.mean_density) < 0.05 * A.mean_density
    # occupancy bits: two runs of the reference sequence with different seeds agree on 98.7 % of the cells of this scene
    # (jitter on the box faces); the sync-free variant is in the same band
    same = float((np.unpackbits(A.density_bitfield.numpy()) == np.unpackbits(B.density_bitfield.numpy())).mean())
    assert same > 0.975, same


def test_scaler_skips_its_gradient_pass_only_when_every_writer_reports():
    """NativeGradScaler._checked_at_source (host logic): the separate non-finite pass over the hand-over buffer is skipped only
    if EVERY adopted parameter reports into THIS scaler's flag and no producer that cannot report (TV term: a torch op) has
    written since zero_grad; a parameter attached to another scaler, or none adopted, keeps the pass."""
    import types
    import torch
    from nerf.optim import NativeGradScaler
    sc, other = NativeGradScaler("cpu"), NativeGradScaler("cpu")

    def param(adopted=True):
        p = types.SimpleNamespace()
        if adopted:
            p._s3d_grad = torch.zeros(4, dtype=torch.float16)
        return p
    a, b, plain = param(), param(), param(adopted=False)
    opt = types.SimpleNamespace(param_groups=[{"params": [a, plain]}, {"params": [b]}])
    assert not sc._checked_at_source(opt)              # nobody attached
    sc.attach(opt)
    assert a._s3d_found_inf is sc._found_inf and not hasattr(plain, "_s3d_found_inf")
    assert sc._checked_at_source(opt) and not other._checked_at_source(opt)
    b._s3d_unchecked = True                            # e.g. GridEncoder.grad_total_variation added to the buffer
    assert not sc._checked_at_source(opt)
    b._s3d_unchecked = False
    b._s3d_found_inf = other._found_inf                # attached elsewhere
    assert not sc._checked_at_source(opt)
    assert not sc._checked_at_source(types.SimpleNamespace(param_groups=[{"params": [plain]}]))  # nothing adopted


def test_sorted_uniform_stream_is_the_order_statistics_of_iid_uniforms():
    """NeRFRenderer._sorted_uniform (cells of the graph-replayed occupancy sweep): ascending, inside [0, 1), and distributed like
    sorted iid uniforms — Kolmogorov-Smirnov distance of the values, uniform spacings (exponential with mean 1/N), and the
    k-th value's mean k/(N+1)"""
    import torch
    from nerf.renderer import NeRFRenderer
    torch.manual_seed(3)
    N = 200000
    u = NeRFRenderer._sorted_uniform(N, torch.device("cpu"))
    assert u.dtype == torch.float64 and u.shape == (N,)
    assert float(u[0]) >= 0.0 and float(u[-1]) < 1.0 and bool((u[1:] >= u[:-1]).all())
    ecdf = torch.arange(1, N + 1, dtype=torch.float64) / N
    assert float((ecdf - u).abs().max()) < 1.63 / N ** 0.5  # KS critical value at alpha = 0.01
    gaps = (u[1:] - u[:-1]) * N
    assert abs(float(gaps.mean()) - 1.0) < 0.01 and abs(float(gaps.var()) - 1.0) < 0.03
    reps = torch.stack([NeRFRenderer._sorted_uniform(999, torch.device("cpu"))[[0, 499, 998]] for _ in range(400)])
    want = torch.tensor([1.0, 500.0, 999.0], dtype=torch.float64) / 1000.0
    assert float((reps.mean(0) - want).abs().max()) < 0.003


def test_tensorf_shrink_model_crops_factors_and_aabb(oracle_wrappers):
    """tensoRF/network.py:273-318: the factors and aabb_train are cropped to the occupied region of the coarsest cascade;
    a point keeps its features (its normalised coordinate moves with the box: same cells, same weights) when the crop
    falls on factor-grid lines"""
    import torch
    from tensoRF import network as trf
    torch.manual_seed(0)
    net = trf.NeRFNetwork(resolution=[33, 33, 33], sigma_rank=[2, 2, 2], color_rank=[3, 3, 3], bound=1, cuda_ray=True, density_thresh=10)
    net.fused_vm = False
    # occupied: the cells whose centres lie in [-0.5, 0.5]^3
    H = net.grid_size
    import raymarching
    idx = torch.arange(H ** 3)
    c = raymarching.morton3D_invert(idx.int()).float()
    centre = (2 * c / (H - 1) - 1) * (1 - 1 / H)
    net.density_grid[0] = ((centre.abs() <= 0.5).all(1)).float() * 100.0
    net.mean_density = float(net.density_grid.clamp(min=0).mean())
    x = (torch.rand(200, 3) - 0.5) * 0.8
    before = net.get_sigma_feat(net._normalize(x)).detach().clone()
    factors = lambda: sum(p.numel() for grp in (net.sigma_mat, net.sigma_vec, net.color_mat, net.color_vec) for p in grp)
    n_before = factors()
    tl, br = net.shrink_model()
    assert all(0 < a < b < 33 for a, b in zip(tl, br)) and net.resolution == [b - a for a, b in zip(tl, br)]
    assert factors() < 0.5 * n_before
    assert tuple(net.sigma_mat[0].shape[-2:]) == (net.resolution[1], net.resolution[0]) and net.sigma_vec[0].shape[-2] == net.resolution[2]
    assert torch.all(net.aabb_train[:3] > -0.6) and torch.all(net.aabb_train[3:] < 0.6) and torch.all(net.aabb_train[3:] >= 0.49)
    # features survive to the accuracy of the box / grid-line mismatch (the crop is rounded to grid lines: the new box is not
    # exactly the span of the kept rows, as in the reference) — a smooth field changes little
    after = net.get_sigma_feat(net._normalize(x)).detach()
    assert torch.isfinite(after).all() and float((after - before).abs().mean()) < 0.6 * float(before.abs().mean())


def test_any_torch_optimizer_step_advances_the_weights_epoch(oracle_wrappers):
    """Fused torch optimizers do not bump Tensor._version, so the fp16-cast caches key on an epoch that every
    optimizer step advances (gridencoder/grid.py)."""
    import gridencoder.grid as gg
    p = torch.nn.Parameter(torch.ones(4))
    opt = torch.optim.SGD([p], lr=0.1)
    p.grad = torch.ones(4)
    before = gg._weights_epoch
    opt.step()
    assert gg._weights_epoch == before + 1


def test_round5_host_logic_of_the_tensorf_step_on_cpu(oracle_wrappers):
    """host-side pieces of the second half of round 5 that need no GPU: the gradient buffers of a factor backward carved out of ONE
    zero fill (16-byte aligned segments, trailing int32 words), the scaler's check pass skipping exactly the gradients whose producer
    raised THIS scaler's flag (and consuming the mark), density_loss_value() == density_loss() without a graph, and the trainers'
    fallbacks on a device without the HIP path (no announced gradient -> autograd L1 term, no fused criterion)"""
    import types
    import torch
    import s3d_hip
    from nerf.optim import NativeGradScaler
    from tensoRF import network as trf
    ts = [torch.empty(1, 3, 5, 5), torch.empty(2, 7), torch.empty(27, 144, dtype=torch.float16)]
    views, words = s3d_hip._zeros_like_many(ts, 4)
    assert [tuple(v.shape) for v in views] == [tuple(t.shape) for t in ts] and all(v.dtype == torch.float32 for v in views)
    assert all(float(v.abs().sum()) == 0 for v in views) and words.dtype == torch.int32 and words.tolist() == [0, 0, 0, 0]
    base = views[0].untyped_storage().data_ptr()
    assert all((v.data_ptr() - base) % 16 == 0 for v in views) and (words.data_ptr() - base) % 16 == 0
    assert len({v.untyped_storage().data_ptr() for v in views}) == 1
    # the scaler's pass over the plain gradients
    sc, other = NativeGradScaler("cpu"), NativeGradScaler("cpu")
    ps = [torch.nn.Parameter(torch.ones(3)) for _ in range(4)]
    for p in ps:
        p.grad = torch.ones(3)
    ps[3].grad[1] = float("inf")
    ps[0]._s3d_grad_checked = sc._found_inf      # its producer reported into this scaler: skipped
    ps[1]._s3d_grad_checked = other._found_inf   # ... into another one: still read
    ps[3]._s3d_grad_checked = sc._found_inf      # a non-finite gradient behind a mark is the producer's to report
    opt = types.SimpleNamespace(grads=lambda: [(None, p, p.grad) for p in ps])
    import nerf.optim as optim_mod
    seen = []

    def cpu_check(g, flag):  # (a stand-in for the HIP per-tensor check: which gradients the pass reads is what is tested)
        seen.append(g)
        if not bool(torch.isfinite(g).all()):
            flag.fill_(1.0)
    real = optim_mod._backend
    optim_mod._backend = types.SimpleNamespace(grads_nonfinite=cpu_check)
    try:
        sc._check_plain(opt, None)
        assert [id(g) for g in seen] == [id(ps[1].grad), id(ps[2].grad)]
        assert float(sc._found_inf) == 0.0 and all("_s3d_grad_checked" not in p.__dict__ for p in ps)
        seen.clear()
        sc._check_plain(opt, None)               # marks are consumed: the next pass reads everything
        assert len(seen) == 4 and float(sc._found_inf) == 1.0
    finally:
        optim_mod._backend = real
    # the L1 term's value
    torch.manual_seed(0)
    net = trf.NeRFNetwork(resolution=[9, 8, 7], sigma_rank=[2, 3, 2], color_rank=[3, 3, 3], bound=1, cuda_ray=True)
    assert abs(float(net.density_loss_value()) - float(net.density_loss())) <= 1e-6 * float(net.density_loss())
    assert not net.density_loss_value().requires_grad and net.density_loss().requires_grad
    st = net.__getstate__()
    assert "_l1_inv" not in st and "_vm_bins" not in st and "_s3d_found_inf" not in st
    # trainers on a device without the HIP path
    from tensoRF.utils import Trainer
    tr = Trainer(net, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-3, fp16=False, update_extra_interval=10 ** 9, native_optim=False)
    assert tr._fused_loss(torch.zeros(8, 3)) is None and net.__dict__.get("_s3d_found_inf") is None
    reg = tr._regularizer()
    assert reg.requires_grad and all("_s3d_l1" not in p.__dict__ for p in net.parameters())
