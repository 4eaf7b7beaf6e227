"""GPU: the training-step variants of nerf/trainer.py on the product path — eager vs HIP-graph replay, the native
optimizer vs torch.optim.Adam + GradScaler, and the data-parallel split-graph step on a 1-rank RCCL group.
Different variants consume the RNG differently (graph capture registers its own philox offsets), so trajectories are
compared as trajectories: same scene, same batches, the loss must fall to the same level."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _setup(n_rays=2048, n_batches=8):
    import bench
    import s3d_hip
    from nerf import network_ff, synthetic as syn
    torch.manual_seed(0)
    model = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    batches, _ = bench.make_batches(n_batches, n_rays, 0, torch.device("cuda"), s3d_hip.RaymarchingBackend,
                                    torch.from_numpy(bits).cuda(), syn.lego_like_boxes(0))
    return model, batches


def _run(trainer, batches, steps):
    losses = []
    for i in range(steps):
        losses.append(trainer.train_step(*batches[i % len(batches)]))
    return torch.stack([l.float().reshape(()) for l in losses]).cpu()


@pytest.mark.parametrize("variant", ["eager-native", "graph-native", "eager-torch"])
def test_training_variants_converge_alike(hip, variant):
    from nerf.trainer import GraphedTrainer, Trainer
    model, batches = _setup()
    if variant == "graph-native":
        tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True)
    else:
        tr = Trainer(model, lr=1e-2, fp16=True, native_optim=(variant == "eager-native"))
    losses = _run(tr, batches, 72)
    assert torch.isfinite(losses).all()
    first, last = float(losses[:8].mean()), float(losses[-8:].mean())
    assert last < 0.5 * first, (variant, first, last)
    assert last < 0.1, (variant, last)  # every variant reaches the same loss level on this scene
    if variant == "graph-native":
        assert tr.n_captures >= 1 and tr.graph is not None
        emb = model.encoder.embeddings
        assert emb.grad is None and (emb._s3d_grad_touched or tr.fuse_table_updates)  # fp16 hand-over / in-backward update active inside the captured step
        assert torch.equal(emb._s3d_half, emb.detach().half())      # the fp16 copy tracks the master weights
    if variant == "eager-torch":
        assert model.encoder.embeddings.grad is not None and not hasattr(model.encoder.embeddings, "_s3d_grad")


def test_second_graphed_trainer_after_the_first_is_gone(hip):
    """Scratch buffers of the binding must not come out of a graph's private memory pool: a second trainer in the same
    process used to replay kernels on scratch that died with the first trainer's graph (illegal memory access)."""
    import gc
    from nerf.trainer import GraphedTrainer
    model, batches = _setup()
    tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True)
    _run(tr, batches, 24)
    assert tr.graph is not None
    del tr, model
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    model, batches = _setup()
    tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True)
    losses = _run(tr, batches, 40)
    torch.cuda.synchronize()
    assert tr.graph is not None and torch.isfinite(losses).all()
    # the step's loss and sample counter are filed by the graph itself: the ring matches the marcher's private counter
    assert int(model.step_counter[(model.local_step - 1) % 16, 0]) > 0
    assert float(losses[-1]) == float(tr.loss_ring[(tr._pushes - 1) % tr.loss_ring.numel()])
    # loss tensors handed out by train_step stay valid for 1,023 further steps: keep 40, train on, compare
    kept = [tr.train_step(*batches[i % len(batches)]) for i in range(40)]
    vals = [float(k) for k in kept]
    _run(tr, batches, 40)
    assert vals == [float(k) for k in kept] and len(set(vals)) > 1


@pytest.mark.parametrize("graphed", [False, True], ids=["eager", "graph"])
def test_overflow_step_is_skipped_with_the_check_made_by_the_gradient_kernels(hip, graphed):
    """single replica: no separate pass over the gradient buffer — the grid / ffmlp backward kernels raise the scaler's flag.
    A batch with non-finite targets must leave every parameter and the Adam step count untouched and halve the scale."""
    from nerf.trainer import GraphedTrainer, Trainer
    model, batches = _setup()
    tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True) if graphed else Trainer(model, lr=1e-2, fp16=True)
    assert tr.scaler._checked_at_source(tr.optimizer)
    _run(tr, batches, 24)
    if graphed:
        assert tr.graph is not None
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in model.parameters()]
    steps0, scale0 = float(tr.optimizer.step_count), tr.scaler.get_scale()
    ro, rd, gt = batches[0]
    bad = gt.clone()
    bad[::3] = float("inf")
    tr.train_step(ro, rd, bad)
    torch.cuda.synchronize()
    assert float(tr.optimizer.step_count) == steps0 and tr.scaler.get_scale() == 0.5 * scale0
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    assert float(tr.scaler._found_inf) == 0.0
    # ... and training goes on: the next clean step updates again, on clean (consumed) gradient buffers
    losses = _run(tr, batches, 8)
    assert float(tr.optimizer.step_count) == steps0 + 8 and torch.isfinite(losses).all()
    assert not all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))


@pytest.mark.parametrize("capture_collectives", [True, False])
def test_data_parallel_graph_step_single_rank(hip, capture_collectives):
    """The multi-GPU step on a 1-rank RCCL group, fp16 flat bucket: (True) ONE graph with the RCCL all-reduce recorded
    inside it, (False) two graphs with an eager all-reduce in between.  Averaging over one rank is the identity, so
    both must train like the single-GPU step."""
    import torch.distributed as dist
    from nerf.trainer import GraphedTrainer
    from parallel import RayShardedDP
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        model, batches = _setup()
        dp = RayShardedDP(force_collective=True)
        tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True, dist=dp, capture_collectives=capture_collectives)
        losses = _run(tr, batches, 56)
        assert tr.graph is not None
        if capture_collectives:
            assert dp.capture_supported(), "RCCL collectives must be capturable on this stack"
            assert tr.graph_opt is None and tr.collectives_in_graph, "the step (all-reduce included) must be ONE graph"
        else:
            assert tr.graph_opt is not None and not tr.collectives_in_graph, "the step must be captured as two graphs"
        assert len(dp.half_grads) == 1 and dp.half_grads[0] is tr.optimizer.flat_half
        assert dp.flat.numel() == 0  # every trainable tensor of this network rides in the fp16 buffer
        assert torch.isfinite(losses).all() and float(losses[-8:].mean()) < 0.5 * float(losses[:8].mean())
    finally:
        # graphs that recorded RCCL kernels go before the communicator they belong to
        tr = dp = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        if created:
            dist.destroy_process_group()


def test_sum_fallback_checks_the_reduced_gradient(hip):
    """SUM + divide reduction (no fused AVG: here a gloo group on GPU tensors): the sum over the ranks can overflow fp16
    although every local gradient is finite, so the REDUCED buffers are checked before any parameter is touched.  Emulated
    on one rank by an all-reduce hook that doubles the buffer (two ranks holding the same large gradient)."""
    import torch.distributed as dist
    from nerf.trainer import Trainer
    from parallel import RayShardedDP
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29579")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="gloo", rank=0, world_size=1)
    try:
        model, batches = _setup()
        model.iter_density = 100
        dp = RayShardedDP(force_collective=True, shard_occupancy=False)
        tr = Trainer(model, lr=1e-2, fp16=True, update_extra_interval=10 ** 9, dist=dp)
        tr.global_step = 1
        assert not dp.fused_avg()
        tr.train_step(*batches[0])  # a clean step first
        before = [p.detach().clone() for p in model.parameters()]
        steps0 = float(tr.optimizer.step_count)
        scale0 = tr.scaler.get_scale()
        orig = dp.allreduce_grads

        def overflowing_sum(scaler=None):
            orig(scaler)
            for h in dp.half_grads:  # what a second rank's equal contribution would do to a large entry
                h[:8] = 60000.0
                h[:8] += h[:8]
        dp.allreduce_grads = overflowing_sum
        tr.train_step(*batches[1])
        dp.allreduce_grads = orig
        assert float(tr.optimizer.step_count) == steps0, "a step whose REDUCED gradient is non-finite must be skipped"
        assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
        assert tr.scaler.get_scale() == scale0 * 0.5
        tr.train_step(*batches[2])
        assert float(tr.optimizer.step_count) == steps0 + 1
    finally:
        if created:
            dist.destroy_process_group()


def test_pipelined_gradient_reduction_single_rank_equals_plain_step(hip):
    """eager trainer, native optimizer, data-parallel layer on a 1-rank RCCL group: the gradient buffer is reduced in pieces
    (cut at parameter boundaries), the overflow flag travels beside them, every parameter is updated once its pieces have
    arrived — with one rank the result must equal the step without the layer bit for bit"""
    import torch.distributed as dist
    from nerf.trainer import Trainer
    from parallel import RayShardedDP
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        res = {}
        for with_dp in (False, True):
            model, batches = _setup()
            model.iter_density = 100
            dp = RayShardedDP(force_collective=True, shard_occupancy=False) if with_dp else None
            if dp is not None:
                dp.chunk_bytes = 1 << 20  # many pieces: the 24 MB table is cut ~24 times
            tr = Trainer(model, lr=1e-2, fp16=True, update_extra_interval=10 ** 9, dist=dp)
            tr.global_step = 1
            torch.manual_seed(7)
            for i in range(4):
                tr.train_step(*batches[i % len(batches)])
            if dp is not None:
                pieces = dp.grad_chunks()
                assert len(pieces) > 10 and pieces[0][1] == 0 and pieces[-1][2] == pieces[-1][0].numel()
            res[with_dp] = [p.detach().clone() for p in model.parameters()]
        for a, b in zip(res[False], res[True]):
            assert torch.equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()


def test_checkpoint_resume_under_graph_replay(hip, tmp_path):
    """train (graph replay, native optimizer) -> full checkpoint in the reference's format -> load into a fresh graphed
    trainer: renders are bit-identical (the fp16 table copies followed the loaded fp32 weights) and training resumes"""
    from nerf.trainer import GraphedTrainer
    model, batches = _setup()
    tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True)
    _run(tr, batches, 40)
    tr.epoch = 1
    path = tr.save_checkpoint(str(tmp_path), full=True)
    ro, rd, _ = batches[0]
    img = tr.render_image(ro, rd)["image"].clone()

    model2, _ = _setup()
    tr2 = GraphedTrainer(model2, 2048, lr=1e-2, fp16=True)
    _run(tr2, batches, 20)  # has a captured graph and its own optimizer state, both must be dropped by the load
    assert tr2.load_checkpoint(path) == ([], [])
    assert tr2.global_step == 40 and model2.mean_count == model.mean_count and tr2.graph is None
    assert float(tr2.optimizer.step_count) == float(tr.optimizer.step_count)
    assert tr2.scaler.get_scale() == tr.scaler.get_scale()
    assert torch.equal(tr2.render_image(ro, rd)["image"], img)
    emb = model2.encoder.embeddings
    assert torch.equal(emb._s3d_half, emb.detach().half())
    before = float(_run(tr, batches, 8).mean())
    after = float(_run(tr2, batches, 8).mean())
    assert tr2.graph is not None and abs(after - before) < 0.5 * before + 1e-3, (before, after)


def test_occupancy_update_replayed_from_its_own_graph(hip):
    """steady state of GraphedTrainer: partial_grid_update_device captured once, replayed every 16 steps, one host read;
    the occupancy it maintains matches the eager reference sequence on the same model (bits differ only where the
    random jitter decides) and training keeps converging"""
    import numpy as np
    from nerf.trainer import GraphedTrainer
    model, batches = _setup()
    tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True)
    _run(tr, batches, 17)       # two full sweeps (steps 0 and 16), eager
    model.iter_density = 16     # steady state from here on
    losses = _run(tr, batches, 80)  # updates at 32 (eager warm-up of the capturable variant), 48 (capture), 64, 80, 96
    assert tr.ues_graph is not None and model.iter_density == 21 and model.mean_count > 0
    assert torch.isfinite(losses).all() and float(losses[-8:].mean()) < 0.1
    bits_graph = np.unpackbits(model.density_bitfield.cpu().numpy())
    assert 0.005 < bits_graph.mean() < 0.5
    # same model, same grid: one more update through each route from the same starting state; two runs of the reference
    # sequence differ by their random jitter — the captured variant must agree with it as well as it agrees with itself
    grid0 = model.density_grid.clone()

    def reference_bits():
        model.density_grid.copy_(grid0)
        with torch.autocast("cuda", dtype=torch.float16):
            model.update_extra_state()
        return np.unpackbits(model.density_bitfield.cpu().numpy())
    bits_ref, bits_ref2 = reference_bits(), reference_bits()
    model.density_grid.copy_(grid0)
    model.local_step = 1
    tr.global_step = 112
    assert tr._maybe_update_extra_state()
    bits_dev = np.unpackbits(model.density_bitfield.cpu().numpy())
    agree_ref, agree_dev = (bits_ref == bits_ref2).mean(), (bits_ref == bits_dev).mean()
    assert agree_dev > agree_ref - 0.01 and agree_dev > 0.9, (agree_ref, agree_dev)
    assert abs(bits_dev.mean() - bits_ref.mean()) < 0.02 * bits_ref.mean() + 1e-3


@pytest.mark.parametrize("net_kind", ["ff", "seal"])
def test_sync_free_inference_loop_renders_the_same_frame(hip, net_kind):
    """nerf/renderer.py:341-367 with the alive-ray count kept on the device (read back every `sync_every` iterations, kernels
    bounded by the device count, rows behind the alive rays skipped through n_valid) against a read-back per iteration and
    against the reference's host boolean-mask compaction: the same frame bit for bit, image and depth."""
    from nerf import network, network_ff, synthetic as syn
    torch.manual_seed(0)
    Net = network_ff.NeRFNetwork if net_kind == "ff" else network.NeRFNetwork
    model = Net(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    grid, bits = syn.lego_like_density_grid(seed=0)
    model.density_grid.copy_(torch.from_numpy(grid))
    model.density_bitfield.copy_(torch.from_numpy(bits))
    for name, p in model.named_parameters():
        if "embeddings" in name:
            p.data.uniform_(-0.5, 0.5)
    model.eval()
    poses = syn.orbit_poses(1, seed=3).cuda()
    r = syn.get_rays(poses, syn.lego_intrinsics(160, 160), 160, 160)
    ro, rd = r["rays_o"].contiguous(), r["rays_d"].contiguous()
    frames = {}
    for tag, compaction, every, scale in (("host", False, 1, 1), ("dev1", True, 1, 1), ("dev4", True, 4, 1), ("dev16", True, 16, 4)):
        model.device_compaction, model.sync_every, model.infer_batch_scale = compaction, every, scale
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out = model.render(ro, rd, bg_color=1, perturb=False, max_steps=1024, T_thresh=1e-4)
        frames[tag] = (out["image"].clone(), out["depth"].clone())
    assert float(frames["host"][0].std()) > 0.01
    for tag in ("dev1", "dev4", "dev16"):
        assert torch.equal(frames[tag][0], frames["host"][0]) and torch.equal(frames[tag][1], frames["host"][1]), tag


def test_long_run_native_fp16_path_converges_like_fp32_adam(hip):
    """6,000 steps of configs[1]'s network from the same initial weights on the same 3.1 M-ray pool: the native path (fp16
    gradient hand-over, exact fixed-point table sums, native Adam + loss scaling, HIP-graph replay, learning-rate schedule read
    from a device word) against torch.optim.Adam on fp32 `.grad`s + torch GradScaler (eager).  PSNR (nerf/utils.py:208-242,
    PSNRMeter) on SIXTEEN held-out 400x400 views of the analytic scene.  The two trajectories are chaotic twins (different
    rounding, different RNG consumption under capture): over 96 initialisations (profiles/r11_psnr_seeds96.json,
    tools/psnr_seeds.py --seeds 96 --steps 6000 --views 16 --hw 400) each arrangement's PSNR scatters with sigma 0.25 - 0.28 dB
    and the paired difference with sigma_d = 0.25 dB around -0.039 dB, 95 % interval [-0.088, +0.010] (half-width 0.049 dB).
    Asserted here: (i) the committed study's interval lies inside +-0.1 dB with a half-width <= 0.08 dB; (ii) a fresh FOUR-seed
    run of the same estimator on this box: both arrangements above 38 dB and the mean difference within
    3 sigma_d / sqrt(4) = 0.37 dB of the study's mean (a 0.1 dB assertion on 4 seeds would fail most runs of identical
    algorithms: its standard error is 0.12 dB)."""
    import argparse
    import json
    import bench
    from nerf import synthetic as syn
    study = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r11_psnr_seeds96.json")))
    lo, hi = study["delta_db_ci95"]
    assert study["seeds"] >= 96 and study["steps"] == 6000 and study["views"].startswith("16 held-out 400x400")
    assert -0.1 <= lo and hi <= 0.1 and (hi - lo) / 2 <= 0.08, study["delta_db_ci95"]
    dev = torch.device("cuda")
    _, bits = syn.lego_like_density_grid(seed=0)
    args = argparse.Namespace(num_rays=4096, seed=0)
    out = bench.long_run_quality(args, dev, hip.RaymarchingBackend, torch.from_numpy(bits).to(dev), syn.lego_like_boxes(0), steps=6000,
                                 seeds=4, n_views=16, hw=400)
    a, b = out["native_fp16_graph"]["psnr_db"], out["torch_adam_fp32_eager"]["psnr_db"]
    assert a >= 38.0 and b >= 38.0, out
    sigma_d = study["delta_db_std"]
    assert out["seeds"] == 4 and abs(out["delta_db"] - study["delta_db"]) <= 3 * sigma_d / 4 ** 0.5, out
    assert out["delta_db_ci95"][0] < out["delta_db"] < out["delta_db_ci95"][1]


_DP2_SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
REPO = os.environ["S3D_REPO"]
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench, s3d_hip
from nerf import network_ff, synthetic as syn
from nerf.trainer import GraphedTrainer
from parallel import RayShardedDP, init_from_env
rank, world, _ = init_from_env("gloo")
torch.cuda.set_device(0)                      # both ranks share the box's one GPU: gloo moves CUDA tensors through the host
torch.manual_seed(100 + rank)                 # replicas start DIFFERENT; register() adopts rank 0's parameters
model = network_ff.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
grid, bits = syn.lego_like_density_grid(seed=0)
batches, _ = bench.make_batches(8, 2048, 0, torch.device("cuda"), s3d_hip.RaymarchingBackend,
                                torch.from_numpy(bits).cuda(), syn.lego_like_boxes(0))
dp = RayShardedDP()
tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True, dist=dp)
losses = []
for i in range(44):                           # 16 eager steps (no sample statistics yet), capture, then replays
    losses.append(float(tr.train_step(*batches[(2 * i + rank) % len(batches)])))   # each rank marches its own rays
torch.cuda.synchronize()
two_graphs = tr.graph is not None and tr.graph_opt is not None and not tr.collectives_in_graph
flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()] +
                 [model.density_grid.reshape(-1), model.density_bitfield.float().reshape(-1)]).cpu()
w = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(w, flat)
same = all(torch.equal(w[0], t) for t in w)
falls = sum(losses[-8:]) < 0.5 * sum(losses[:8])
print(f"RANK{rank} two_graphs={two_graphs} same={same} falls={falls} finite={bool(torch.isfinite(flat).all())} "
      f"fused_avg={dp.fused_avg()} replays={44 - 16 - tr.n_captures}")
dist.destroy_process_group()
'''


def test_two_graph_dp_step_on_two_gloo_ranks(hip, tmp_path):
    """GraphedTrainer's two-graph data-parallel step end to end on TWO ranks (gloo, both on this box's one GPU): forward +
    backward graph, eager SUM all-reduce of the fp16 gradient buffer + divide, check + Adam graph; each rank marches its own
    rays; after 44 steps (occupancy updates included) the replicas' parameters and occupancy grids are bit-identical."""
    import subprocess
    script = tmp_path / "dp2.py"
    script.write_text(_DP2_SCRIPT)
    env = dict(os.environ, S3D_REPO=REPO, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    for r in (0, 1):
        line = [l for l in res.stdout.splitlines() if l.startswith(f"RANK{r} ")]
        assert line and "two_graphs=True same=True falls=True finite=True fused_avg=False" in line[0], res.stdout + res.stderr[-1500:]


@pytest.mark.parametrize("hidden", [128, 16])
def test_network_with_library_path_widths_trains_and_renders(hip, hidden):
    """FFMLP accepts hidden 16 / 128 / 256 through the layer-by-layer path, which has no level-major / n_valid /
    head routes — a NeRFNetwork of such a width must take the reference's op sequence instead of crashing in train_step,
    update_extra_state or render."""
    import bench
    import s3d_hip
    from nerf import network_ff, synthetic as syn
    from nerf.trainer import Trainer
    torch.manual_seed(0)
    model = network_ff.NeRFNetwork(hidden_dim=hidden, hidden_dim_color=hidden, bound=1, cuda_ray=True, density_scale=1,
                                   min_near=0.2, density_thresh=10).cuda()
    assert not model._fused_mlps()
    grid, bits = syn.lego_like_density_grid(seed=0)
    batches, poses = bench.make_batches(4, 1024, 0, torch.device("cuda"), s3d_hip.RaymarchingBackend,
                                        torch.from_numpy(bits).cuda(), syn.lego_like_boxes(0))
    tr = Trainer(model, lr=1e-2, fp16=True, update_extra_interval=16)
    losses = _run(tr, batches, 40)  # (covers two occupancy updates)
    assert torch.isfinite(losses).all() and float(losses[-4:].mean()) < float(losses[:4].mean())
    r = syn.get_rays(poses[:1].cuda(), syn.lego_intrinsics(32, 32), 32, 32)
    img = tr.render_image(r["rays_o"].contiguous(), r["rays_d"].contiguous())["image"]
    assert img.shape[-1] == 3 and torch.isfinite(img).all()


@pytest.mark.parametrize("graphed", [False, True], ids=["eager", "graph"])
def test_step_with_the_one_launch_criterion_equals_the_three_launch_step_bit_for_bit(hip, graphed):
    """Trainer.fused_losses: compositing, background + MSE and the compositing backward as one launch
    (raymarching.composite_rays_train_loss) against composite_rays_train -> _BgMse -> backward — same steps, same RNG, every
    loss and every parameter afterwards identical; the graph holds two launches fewer"""
    from nerf.trainer import GraphedTrainer, Trainer
    res, calls = {}, {True: 0, False: 0}
    R = hip.RaymarchingBackend
    inner = R.composite_rays_train_loss
    for fused in (True, False):
        model, batches = _setup()
        model.iter_density = 100
        tr = (GraphedTrainer(model, 2048, lr=1e-2, fp16=True, update_extra_interval=10 ** 9) if graphed
              else Trainer(model, lr=1e-2, fp16=True, update_extra_interval=10 ** 9))
        tr.fused_losses = fused
        tr.global_step = 1
        torch.manual_seed(7)

        def counted(*a, **k):
            calls[fused] += 1
            return inner(*a, **k)
        R.composite_rays_train_loss = staticmethod(counted)
        try:
            losses = _run(tr, batches, 6)
        finally:
            R.composite_rays_train_loss = staticmethod(inner)
        assert torch.isfinite(losses).all()
        res[fused] = (losses, [p.detach().clone() for p in model.parameters()])
    assert calls[True] > 0 and calls[False] == 0, calls
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("graphed", [False, True])
def test_table_update_inside_the_backward_equals_the_separate_update_bit_for_bit(hip, graphed):
    """The hash table's Adam applied in the grid backward's accumulate kernel (s3d_grid_encode_backward_adam; single replica)
    against the separate update (gradient table written, s3d_adam_step_multi): same batches, same initial weights -> the same
    master table, moments, fp16 copy and MLP weights after 40 steps, bit for bit (the in-backward update uses the binary16
    value the gradient table would hold), and the same losses."""
    from nerf.trainer import GraphedTrainer, Trainer
    res = {}
    for fuse in (True, False):
        model, batches = _setup()
        tr = GraphedTrainer(model, 2048, lr=1e-2, fp16=True) if graphed else Trainer(model, lr=1e-2, fp16=True)
        tr.fuse_table_updates = fuse
        if graphed:
            tr.noise_key = 1234
        torch.manual_seed(7)
        losses = _run(tr, batches, 40)
        emb = model.encoder.embeddings
        st = tr.optimizer.state[emb]
        res[fuse] = (losses, emb.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), emb._s3d_half.clone(),
                     [p.detach().clone() for p in model.parameters()], float(tr.optimizer.step_count))
        if fuse:
            assert not emb._s3d_grad_touched and float(emb._s3d_grad.abs().max()) == 0.0, "the gradient table must stay untouched"
        else:
            assert emb._s3d_grad_touched
    a, b = res[True], res[False]
    assert a[6] == b[6] == 40.0
    assert torch.equal(a[0], b[0]), (a[0] - b[0]).abs().max()
    for k in (1, 2, 3, 4):
        assert torch.equal(a[k], b[k]), k
    assert all(torch.equal(x, y) for x, y in zip(a[5], b[5]))
    assert torch.equal(a[4], a[1].half())
