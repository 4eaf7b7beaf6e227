#!/usr/bin/env python3
"""bench.py — hot-path benchmark for the MI355X build (contract: see the task statement / DESIGN.md §Measurement).

A "step" is one NGP training step of BASELINE.json configs[1] on synthetic Lego-shaped input:
  4,096 rays of an 800x800 Blender-Lego camera -> near/far -> march_rays_train (occupancy bitfield) ->
  hash-grid encode (L=16, F=2, T=2^19) -> fused MLPs (ffmlp 32-64-64?-16 / 32-64-64-3, fp16 MFMA) + SH-4 ->
  composite_rays_train -> MSE -> backward (composite bwd, ffmlp bwd, grid scatter) -> Adam (fp16 autocast +
  GradScaler), `update_extra_state` every 16 steps.  Ray batches and targets are resident in HBM before timing.
`value` = real marched samples (sum of step_counter[:,0] over the timed steps and over ranks) / wall time.

Multi-GPU (`python -m torch.distributed.run ... bench.py --gpus N`): weak scaling, every rank trains on its own
4,096-ray batches; one gradient all-reduce (RCCL) per step over the optimizer's flat fp16 buffer, recorded INSIDE the step's
HIP graph when the process group supports capture (two graphs with an eager collective between them otherwise).
`python bench.py --gpus N` without torchrun re-launches itself as N ranks and refuses (exit 2) when fewer GPUs are visible;
rank 0 checks that N ranks took part in the gradient all-reduce before it prints the line.

Extra objects on the JSON line:
  roofline      dominant hot kernel of the timed region, timed with HIP events on the launch stream
  cpu_baseline  the same step on the host cores through the CPU oracle ("port"), bounded sample, rank 0, N=1 only
"""
import argparse
import json
import math
import os
import re
import sys
import time

# (host-side only: OpenMP teams of the CPU baseline wait passively — two runtimes with spinning teams, torch's and the
#  oracle's, starve each other on a many-core host.  Has to be in the environment before libgomp initialises.)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np  # noqa: E402
import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "seal-3d_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--num_rays", type=int, default=4096)
    ap.add_argument("--pretrain", type=int, default=384, help="untimed setup steps that converge the occupancy grid")
    ap.add_argument("--net", choices=["ff", "seal"], default="ff", help="ff: nerf/network_ff (configs[1]); seal: two-encoder nn.Linear net")
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--force_dp", action="store_true", help="run the data-parallel path (flat-bucket all-reduce, split graphs) on 1 GPU")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_render", action="store_true")
    ap.add_argument("--save_model", default=None, help="write the trained model's state_dict here after the timed region "
                    "(tools/profile_render.sh renders THIS model under rocprofv3)")
    ap.add_argument("--infer_batch_scale", type=int, default=4, help="inference samples/ray/iteration multiplier (1 = reference heuristic)")
    ap.add_argument("--cpu_steps", type=int, default=5, help="timed CPU-baseline steps (median), after --cpu_warmup warm-ups (BASELINE.md §3: 2 + 5)")
    ap.add_argument("--cpu_warmup", type=int, default=2)
    ap.add_argument("--cpu_rays", type=int, default=4096, help="rays per CPU-baseline step (BASELINE.md §3 (ii): 4,096)")
    ap.add_argument("--no_cpu_render", action="store_true", help="skip the 64x64 CPU renders (cuda_ray on and off) of BASELINE.md §3 (i)")
    ap.add_argument("--no_seal", action="store_true", help="skip the configs[2] (Seal bbox distillation) section")
    ap.add_argument("--no_tensorf", action="store_true", help="skip the configs[4] (TensoRF VM-48 training step) section")
    ap.add_argument("--no_long_run", action="store_true", help="skip the 16 x 2 x 6,000-step convergence comparison (psnr.long_run)")
    ap.add_argument("--long_run_steps", type=int, default=6000)
    ap.add_argument("--long_run_seeds", type=int, default=16, help="initialisations of the long-run comparison (per-seed sigma of the paired difference 0.25 dB: 16 seeds = +-0.13 dB at 95 %%)")
    ap.add_argument("--long_run_views", type=int, default=16, help="held-out views of the long-run PSNR (400x400 each)")
    ap.add_argument("--seal_teacher_steps", type=int, default=256)
    ap.add_argument("--seal_point_step", type=float, default=0.005, help="pretraining_local_point_step (readme.md:109)")
    ap.add_argument("--seal_surrounding_step", type=float, default=0.01, help="pretraining_surrounding_point_step (main_SealNeRF.py:98; <= 0: off)")
    ap.add_argument("--seal_proxy_poses", type=int, default=4, help="poses per rank whose frames proxy_dataset renders")
    ap.add_argument("--seal_frame", type=int, default=800, help="frame size of the proxied dataset")
    ap.add_argument("--seal_iters", type=int, default=30000, help="--iters of main_SealNeRF.py (LambdaLR horizon)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None, help="process-group backend (default: nccl = RCCL)")
    ap.add_argument("--rendezvous_only", action="store_true",
                    help="launch check: the ranks rendezvous, count themselves with an all-reduce, rank 0 prints the count; no measurement")
    return ap.parse_args()


# ----------------------------------------------------------------------------- synthetic data
def analytic_targets(rays_o, rays_d, scene_bits, boxes, R):
    """ground-truth colours of the box scene for a ray batch, composited with the build's own kernels"""
    from nerf import synthetic as syn
    N = rays_o.shape[0]
    dev = rays_o.device
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=dev)
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    R.near_far_from_aabb(rays_o, rays_d, aabb, N, 0.2, nears, fars)
    M = N * 256
    xyzs, dirs, deltas = (torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev))
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    R.march_rays_train(rays_o, rays_d, scene_bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                       torch.zeros(N, device=dev))
    lo, hi = boxes
    sig = syn.box_density(xyzs, lo, hi, sigma=60.0)
    rgb = (0.5 + 0.5 * torch.sin(xyzs * 9.0 + torch.tensor([0.0, 2.0, 4.0], device=dev))).contiguous()
    ws, dp, im = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    R.composite_rays_train_forward(sig.contiguous(), rgb, deltas, rays, M, N, 1e-4, ws, dp, im)
    return im + (1 - ws).unsqueeze(-1)  # white background


def make_batches(n_batches, num_rays, seed, dev, R, scene_bits, boxes):
    from nerf import synthetic as syn
    poses = syn.orbit_poses(100, seed=0)  # shared camera set (torch.manual_seed(0) convention of SURVEY §8d)
    g = torch.Generator().manual_seed(1000 + seed)
    out = []
    for b in range(n_batches):
        k = int(torch.randint(0, poses.shape[0], (1,), generator=g))
        r = syn.get_rays(poses[k:k + 1], syn.lego_intrinsics(), 800, 800, N=num_rays, generator=g)
        ro, rd = r["rays_o"][0].contiguous().to(dev), r["rays_d"][0].contiguous().to(dev)
        out.append((ro, rd, analytic_targets(ro, rd, scene_bits, boxes, R)))
    return out, poses


# ----------------------------------------------------------------------------- kernel timing hooks
class KernelTimers:
    """HIP-event timing of individual native calls on the launch stream (torch's current stream)."""

    def __init__(self, backend_cls, names):
        self.cls, self.names, self.orig, self.events = backend_cls, names, {}, {n: [] for n in names}
        self.meta = {n: [] for n in names}

    def install(self, meta_fn, queue_ahead=False):
        """`queue_ahead` (only outside the timed region): a ~60 us GPU spin is enqueued in front of the start event, so the
        call's kernels are all queued by the time the first one may start and run back to back as they do inside the replayed
        graph — without it the bracket also counts the host's launch latency between the kernels of one call (the binned
        grid backward is three launches: 167 us bracketed against 150 us of kernel time in the rocprofv3 trace)"""
        for n in self.names:
            f = getattr(self.cls, n)
            self.orig[n] = f

            def wrapped(*a, __f=f, __n=n, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if queue_ahead:
                    torch.cuda._sleep(150000 if queue_ahead is True else int(queue_ahead))
                s.record()
                r = __f(*a, **kw)
                e.record()
                self.events[__n].append((s, e))
                nv = kw.get("n_valid")  # padded batch: the kernel works on round_up(*n_valid, 128) of the B rows
                self.meta[__n].append((meta_fn(__n, a), None if nv is None else nv.reshape(-1)[:1].clone()))
                return r
            setattr(self.cls, n, staticmethod(wrapped))

    def remove(self):
        for n, f in self.orig.items():
            setattr(self.cls, n, staticmethod(f))

    def reset(self):
        for n in self.names:
            self.events[n].clear()
            self.meta[n].clear()

    def summary(self):
        out = {}
        for n in self.names:
            if not self.events[n]:
                continue
            ms = [s.elapsed_time(e) for s, e in self.events[n]]
            units = [B if nv is None else min(B, (max(int(nv.item()), 0) + 127) // 128 * 128) for B, nv in self.meta[n]]
            out[n] = dict(calls=len(ms), total_ms=float(sum(ms)), avg_us=float(np.mean(ms) * 1e3),
                          units=float(np.mean(units)))
        return out


def source_digest(rel="seal-3d_amd/csrc/gridencoder.hip"):
    """sha256 (first 16 hex digits) of a kernel source file: profiles/rNN_timed_region.md records the one it was measured on"""
    import hashlib
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def profile_traffic(op):
    """HBM bytes per launch of the dominant op from the newest committed rocprofv3 PMC summary (profiles/rNN_timed_region.md:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, gfx950 2x fetch correction applied,
    tools/profile_bench.sh).  PMC counters cannot be collected from inside this process.  The summary names the sha256 of the
    grid-encoder source it was measured on: a summary of OTHER kernels than the ones this run launches is refused (None)
    rather than quoted — as is a summary without the digest line."""
    import glob
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(f for f in glob.glob(os.path.join(here, "profiles", "r*_timed_region.md"))
                   if re.match(r"r\d+_timed_region\.md$", os.path.basename(f)))  # (rNNplain_*: the A/B run with the update separate)
    if not files:
        return None, None
    text = open(files[-1]).read()
    m = re.search(r"gridencoder\.hip sha256:([0-9a-f]{16})", text)
    if not m or m.group(1) != source_digest():
        return None, f"{os.path.relpath(files[-1], here)} is stale (measured on another gridencoder.hip)"
    kernels = ("k_grid_forward_pair",) if op == "grid_encode_forward" else ("k_bin_scatter6", "k_bin_accumulate6")  # (both backward ops)
    total, seen = 0.0, set()
    for line in text.splitlines():
        m = re.match(r"\| `([A-Za-z0-9_]+)", line)
        if not m or m.group(1) not in kernels or m.group(1) in seen:
            continue
        cols = [c.strip() for c in line.strip().strip("|").split("|")]
        try:
            total += (float(cols[5]) + float(cols[6])) * 1024.0  # FETCH x2 KiB + WRITE KiB
            seen.add(m.group(1))
        except (ValueError, IndexError):
            return None, None
    if len(seen) != len(kernels):
        return None, None
    return total, os.path.relpath(files[-1], here)


def grid_meta(name, args):
    # grid_encode_forward(inputs, embeddings, offsets, outputs, B, ...) / backward(grad, inputs, embeddings, offsets, ge, B, ...)
    return args[4] if name == "grid_encode_forward" else args[5]  # (backward_adam: the same leading arguments as backward)


# ----------------------------------------------------------------------------- CPU baseline (oracle port)
def cpu_baseline(args, num_rays):
    """the same training step (two-pass marching, hash encode, nn.Linear MLPs = the reference's `--ff`-off network,
    composite, backward, Adam) on the host cores, native ops by the CPU oracle (OpenMP).  BASELINE.md §3 (ii): 4,096 rays per
    step, 2 warm-ups + median of 5.  The step is a chain of small parallel regions (torch's intra-op pool for the MLPs and
    Adam, the oracle's OpenMP teams for the native ops): on a many-core host the thread count that wins is NOT "all of them"
    (round 3 ran both pools at 256 threads and measured 10.7 s per 1,024-ray step — two spinning teams of 256 taking turns),
    so the step is timed at {16, 64, all} threads (both pools at the same count, passive waiting) and the best is reported
    with its thread count; a setting whose first step takes more than 3 s is cut to one timed step."""
    from oracle import oracle_backend as ob
    import raymarching.raymarching as rm
    import gridencoder.grid as gg
    import shencoder.sphere_harmonics as sh
    from nerf import network, synthetic as syn
    from nerf.trainer import Trainer
    ob.build()
    cores = os.cpu_count() or 1
    sweep = sorted({t for t in (16, 64, cores) if t <= cores} or {cores})
    saved = (rm._backend, gg._backend, sh._backend)
    rm._backend, gg._backend, sh._backend = ob.RaymarchingBackend, ob.GridBackend, ob.SHBackend
    try:
        grid, bits = syn.lego_like_density_grid(seed=0)
        boxes = syn.lego_like_boxes(0)
        batches, poses = make_batches(2, num_rays, 0, "cpu", ob.RaymarchingBackend, torch.from_numpy(bits), boxes)

        def fresh():
            torch.manual_seed(0)
            net = network.NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
            net.density_grid.copy_(torch.from_numpy(grid))
            net.density_bitfield.copy_(torch.from_numpy(bits))
            net.iter_density = 100  # skip the full-sweep grid update: the sample is the steady-state step
            tr = Trainer(net, fp16=False, update_extra_interval=10 ** 9)
            tr.global_step = 1
            return net, tr

        def one_step(net, tr, i):
            t0 = time.perf_counter()
            tr.train_step(*batches[i % 2])
            dt = time.perf_counter() - t0
            return int(net.step_counter[(net.local_step - 1) % 16, 0]), dt
        by_threads, t_all, samples = {}, time.perf_counter(), 0
        for t in sweep:
            torch.set_num_threads(t)
            ob.set_threads(t)
            net, tr = fresh()
            n, dt = one_step(net, tr, 0)  # first warm-up, also the probe
            slow = dt > 3.0
            for i in range(1, 1 if slow else args.cpu_warmup):
                one_step(net, tr, i)
            rates = []
            for i in range(1 if slow else args.cpu_steps):
                n, dt = one_step(net, tr, i)
                rates.append(n / dt)
                samples += n
            by_threads[t] = {"samples_per_s": float(np.median(rates)), "timed_steps": len(rates),
                             "ms_per_step": float(np.median([1e3 * n / r for r in rates]))}
        dt_all = time.perf_counter() - t_all
        best = max(by_threads, key=lambda t: by_threads[t]["samples_per_s"])
        torch.set_num_threads(best)
        ob.set_threads(best)
        net, tr = fresh()
        render = {}
        if not args.no_cpu_render:  # a 64x64 full render (4,096 rays), `cuda_ray` on (inference loop) ...
            r = syn.get_rays(poses[:1], syn.lego_intrinsics(64, 64), 64, 64)
            net.device_compaction = False
            t0 = time.perf_counter()
            tr.render_image(r["rays_o"].contiguous(), r["rays_d"].contiguous())
            render["render_64x64_rays_per_s"] = 4096 / (time.perf_counter() - t0)
            # ... and off: NeRFRenderer.run, 512 + 128 samples per ray (main_SealNeRF.py:47), staged like the reference's test loop
            net.cuda_ray = False
            net.eval()
            try:
                t0 = time.perf_counter()
                with torch.no_grad():
                    net.render(r["rays_o"].contiguous(), r["rays_d"].contiguous(), staged=True, max_ray_batch=1024, bg_color=1,
                               perturb=False, num_steps=512, upsample_steps=128)
                render["render_64x64_rays_per_s_cuda_ray_off"] = 4096 / (time.perf_counter() - t0)
            finally:
                net.cuda_ray = True
    finally:
        rm._backend, gg._backend, sh._backend = saved
        torch.set_num_threads(cores)
    model_name = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model_name = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": by_threads[best]["samples_per_s"], "unit": "samples/s", "cores": best, "host_cores": cores, "cpu_model": model_name,
            "kind": "port", "thread_sweep": {str(t): v for t, v in by_threads.items()}, **render,
            "sample": f"median of {args.cpu_steps} training steps (after {args.cpu_warmup} warm-ups) x {num_rays} rays per thread setting "
                      f"{sweep} ({samples} samples, {dt_all:.1f} s in all), best setting reported; same synthetic scene, fp32, two-encoder "
                      "nn.Linear network (the reference's --ff-off path), native ops = CPU oracle + OpenMP; the 64x64 renders use the "
                      "best setting"}


# ----------------------------------------------------------------------------- PSNR of the HIP render against the oracle render
def psnr_vs_oracle(model, make_model, poses, scene_bits, boxes, dev, R):
    """north_star: PSNR within 0.1 dB of the reference path.  One 64x64 frame of the TRAINED model rendered (a) by the HIP
    path and (b) by the CPU oracle under the same Python (same weights, same precision: fp32 tables, fp16 MLPs, no autocast);
    PSNR of each against the analytic target with the reference's PSNRMeter formula (nerf/utils.py:226-233)."""
    from oracle import oracle_backend as ob
    import raymarching.raymarching as rm
    import gridencoder.grid as gg
    import shencoder.sphere_harmonics as sh
    import ffmlp.ffmlp as ff
    from nerf import synthetic as syn
    from nerf.trainer import psnr
    ob.build()
    r = syn.get_rays(poses[:1].to(dev), syn.lego_intrinsics(64, 64), 64, 64)
    ro, rd = r["rays_o"].contiguous(), r["rays_d"].contiguous()
    target = analytic_targets(ro[0].contiguous(), rd[0].contiguous(), scene_bits, boxes, R)
    kw = dict(bg_color=1, perturb=False, max_steps=1024, T_thresh=1e-4, dt_gamma=0)
    model.eval()
    with torch.no_grad():
        hip = model.render(ro, rd, **kw)["image"][0]
    cpu_model = make_model()  # a fresh CPU instance with the trained state (no GPU-side optimizer attachments)
    cpu_model.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    cpu_model.eval()
    cpu_model.device_compaction = False
    saved = (rm._backend, gg._backend, sh._backend, ff._backend)
    rm._backend, gg._backend, sh._backend, ff._backend = ob.RaymarchingBackend, ob.GridBackend, ob.SHBackend, ob.FFMLPBackend
    try:
        ob.set_threads(os.cpu_count() or 1)
        with torch.no_grad():
            ora = cpu_model.render(ro.cpu(), rd.cpu(), **kw)["image"][0]
    finally:
        rm._backend, gg._backend, sh._backend, ff._backend = saved
    model.train()
    p_hip, p_ora = psnr(hip.cpu(), target.cpu()), psnr(ora, target.cpu())
    return {"hip_vs_target_db": p_hip, "oracle_vs_target_db": p_ora, "delta_db": abs(p_hip - p_ora),
            "hip_vs_oracle_render_db": psnr(hip.cpu(), ora), "within_0p1_db": bool(abs(p_hip - p_ora) <= 0.1),
            "frame": "64x64, trained weights, fp32 tables + fp16 MLPs on both sides, PSNRMeter formula"}


# ----------------------------------------------------------------------------- configs[2]: Seal bbox distillation
SEAL_BBOX = {"type": "bbox", "raw": [[x, y, z] for x in (-0.2, 0.2) for y in (0.0, 0.3) for z in (-0.2, 0.2)],
             "transform": [[1, 0, 0, 0.3], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], "scale": [1, 1, 1], "boundType": "both"}


def seal_section(args, dev, batches, note=lambda m: None, make_dp=None, reps=None, eager=False, net_kw=None):
    """BASELINE configs[2] (1 GPU) / configs[3] (`--gpus N`: SURVEY §8e), in the order main_SealNeRF.py runs them — teacher = the
    two-encoder NGP net trained on the synthetic scene and left in eval mode (:210), student = its copy, bbox edit translate
    (0.3, 0, 0), LambdaLR 0.1 ** (iter / iters) stepped every step (:283-300);
      init_pretraining  local lattice at pretraining_local_point_step 0.005 (~7.7e5 points) + the surrounding lattice at the
                        CLI defaults (step 0.01, bounds grown by 0.1; :98-103), one chunk each (batch 6,144,000);
      pretraining       epochs over both parts, MLPs frozen;
      proxy_dataset     every training pose's 800x800 frame rendered ONCE by the eval-mode teacher through the proxy
                        (SealNeRF/provider.py:19-70, called by train() :270-273) -> colour + depth targets;
      fine-tuning       4,096-ray batches per rank whose targets are gathered from those frames (`skip_proxy`): the student's
                        step alone — `seal_train_ms_per_step`.  Also timed: the step with the per-step teacher render the
                        reference's GUI loop does (train_gui -> proxy_truth, eval-mode teacher: `*_online_proxy`), and this
                        build's one-march variant of it (teacher's training branch replayed from a HIP graph: `*_one_march`).
    With N ranks (`make_dp()` -> a parallel.RayShardedDP per trainer): every chunk of pretraining points is sharded over the
    ranks (SealNeRF/trainer.py:404-413; MLPs frozen: table gradients only), each rank proxies its own poses and fine-tunes on
    its OWN rays (weak scaling), and both tables' and both MLPs' gradients travel in the one all-reduce per step.  Every timing
    is bracketed by a barrier and is the MAX over the ranks; rates are whole-job.
    `eager`: the eager trainers instead of the graph-replayed ones (the only choice without a GPU: tests/); `reps`, `net_kw`:
    repetition counts / network arguments of a reduced run (tests/)."""
    import torch.distributed as dist
    from nerf import network, synthetic as syn
    from nerf.trainer import GraphedTrainer, Trainer
    from sealnerf import GraphedSealTrainer, SealBBoxMapper, SealDataset, SealTrainer, make_student, make_teacher
    reps = dict(dict(pretrain=8, proxy=8, warm=40, step=32, proxy_graph=16, allreduce=8, online=8), **(reps or {}))
    dpt, dps = (make_dp(), make_dp()) if make_dp is not None else (None, None)
    world = dps.world if dps is not None else 1
    rank = dps.rank if dps is not None else 0
    multi = world > 1 and dist.is_initialized()
    kw = dict(dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10), **(net_kw or {}))
    torch.manual_seed(args.seed + 17)
    teacher = make_teacher(network.NeRFNetwork, **kw).to(dev)
    ttr = (Trainer(teacher, lr=1e-2, fp16=True, update_extra_interval=16, dist=dpt) if eager else
           GraphedTrainer(teacher, args.num_rays, lr=1e-2, fp16=True, update_extra_interval=16, dist=dpt))
    for i in range(args.seal_teacher_steps):
        ttr.train_step(*batches[i % len(batches)])
    del ttr
    teacher.train(False)  # main_SealNeRF.py:210: every teacher render below takes run_cuda's inference loop
    student = make_student(network.NeRFNetwork, **kw).to(dev)
    student.load_state_dict(teacher.state_dict())
    student.mean_count, student.mean_density, student.iter_density = teacher.mean_count, teacher.mean_density, teacher.iter_density
    mapper = SealBBoxMapper(SEAL_BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    note("seal: teacher trained; pretraining")

    def scheduler(opt):
        return torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / args.seal_iters, 1))
    tr = (SealTrainer(student, teacher, lr=1e-2, fp16=True, update_extra_interval=16, dist=dps, lr_scheduler=scheduler) if eager else
          GraphedSealTrainer(student, teacher, args.num_rays, lr=1e-2, fp16=True, update_extra_interval=16, dist=dps, lr_scheduler=scheduler))
    n_local = tr.init_pretraining(batch_size=6144000, lr=0.05, local_point_step=args.seal_point_step,
                                  surrounding_point_step=args.seal_surrounding_step, surrounding_bounds_extend=0.1, global_point_step=-1)
    n_points = {k: int(v["points"].shape[0]) for k, v in tr.pretraining_data.items()}

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def sync_time(fn, n):
        sync()
        if multi:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        sync()
        if multi:
            dist.barrier()
        dt = torch.tensor([(time.perf_counter() - t0) / n], dtype=torch.float64, device=dev)
        if multi:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item())

    def total(t):  # whole-job count: the sum over the ranks
        t = t.to(torch.float64)
        if multi:
            dist.all_reduce(t)
        return float(t.item())
    tr.pretrain_one_epoch()
    l0 = float(tr.last_pretrain_losses[0])  # (the local part's first chunk: the edit region, where student != teacher's proxy)
    tr.pretrain_one_epoch()  # (second epoch: the chunks' graphs are captured)
    ep = sync_time(lambda i: tr.pretrain_one_epoch(), reps["pretrain"])
    tr.pretrain_one_epoch()
    l1 = float(tr.last_pretrain_losses[0])
    note(f"seal: pretraining timed ({ep * 1e3:.2f} ms/epoch); proxy dataset")
    # ---- proxy_dataset: this rank's poses, whole frames, eval-mode teacher through the proxy
    F_ = args.seal_frame
    n_frames = max(1, args.seal_proxy_poses)
    poses = syn.orbit_poses(100, seed=0)[rank * n_frames:(rank + 1) * n_frames].to(dev)
    ds = SealDataset(poses, syn.lego_intrinsics(F_, F_), F_, F_, num_rays=args.num_rays, device=dev, render_kwargs=tr.render_kwargs, fp16=True)
    ds.proxy_dataset(teacher)  # (warm-up: allocator pools, lazy initialisations)
    pd = sync_time(lambda i: ds.proxy_dataset(teacher), 1)
    proxy = sync_time(lambda i: tr.proxy_truth(batches[i % len(batches)][0], batches[i % len(batches)][1]), reps["proxy"])
    note(f"seal: proxy dataset {n_frames} frame(s) in {pd * 1e3:.1f} ms; fine-tuning")
    g = torch.Generator().manual_seed(args.seed + 23 + rank)
    nb = min(16, max(len(batches), 2))
    coll = []
    for i in range(nb):
        b = ds.collate([i % n_frames], generator=g)
        coll.append((b["rays_o"][0].contiguous(), b["rays_d"][0].contiguous(), b["images"][0].contiguous(), b["depths"][0, :, 0].contiguous()))
    for i in range(reps["warm"]):  # fine-tuning warm-up: 16 eager steps (sample statistics), capture, replays
        tr.train_step(*coll[i % nb])
    samples = torch.zeros(1, dtype=torch.int64, device=dev)

    def ft(i):
        tr.train_step(*coll[i % nb])
        samples.add_(student.step_counter[(student.local_step - 1) % 16, 0].long())
    step = sync_time(ft, reps["step"])

    def ft_collate(i):
        b = ds.collate([i % n_frames], generator=g)
        tr.train_step(b["rays_o"][0], b["rays_d"][0], b["images"][0], b["depths"][0, :, 0])
    step_collate = sync_time(ft_collate, reps["step"])
    # ---- the GUI loop's per-step teacher render (train_gui -> train_step -> proxy_truth), eval-mode teacher
    samples_on = torch.zeros(1, dtype=torch.int64, device=dev)

    def ft_online(i):
        tr.train_step(batches[i % len(batches)][0], batches[i % len(batches)][1])
        samples_on.add_(student.step_counter[(student.local_step - 1) % 16, 0].long())
    ft_online(0)
    step_online = sync_time(ft_online, reps["online"])
    # ---- this build's one-march variant: the teacher's training branch (force_all_rays), replayed from its own graph
    tr.online_proxy_mode = "train"
    for i in range(3):
        tr.train_step(batches[i % len(batches)][0], batches[i % len(batches)][1])
    samples_om = torch.zeros(1, dtype=torch.int64, device=dev)

    def ft_one_march(i):
        tr.train_step(batches[i % len(batches)][0], batches[i % len(batches)][1])
        samples_om.add_(student.step_counter[(student.local_step - 1) % 16, 0].long())
    step_one_march = sync_time(ft_one_march, reps["step"])
    proxy_graph = None
    if getattr(tr, "proxy_graph", None) is not None:
        def proxy_replay(i):  # the proxy render as that step runs it: rays staged, its own HIP graph replayed
            b = batches[i % len(batches)]
            torch._foreach_copy_([tr.s_ro, tr.s_rd], [b[0].reshape(-1, 3), b[1].reshape(-1, 3)])
            tr._proxy_replay()
        proxy_graph = sync_time(proxy_replay, reps["proxy_graph"])
    tr.online_proxy_mode = None
    n_samples, n_on, n_om = total(samples), total(samples_on), total(samples_om)
    n_epoch = sum(n_points.values())
    out = {"workload": ("configs[2]: lego_bbox-shaped edit (bbox translate 0.3), teacher+student two-encoder NGP, "
                        f"pretraining_local_point_step={args.seal_point_step:g} + surrounding {args.seal_surrounding_step:g}, "
                        f"{args.num_rays} rays/step, targets from the proxied dataset, 1 GPU, HIP-graph replay") if world == 1 and not eager else
                       ("configs[3]: lego_bbox-shaped edit, teacher+student two-encoder NGP distillation, pretraining points sharded "
                        f"over {world} rank(s), {args.num_rays} rays/step/rank, one gradient all-reduce per step"),
           "local_points": int(n_local), "pretrain_points": n_points,
           "seal_pretrain_points_per_s": n_epoch / ep, "pretrain_ms_per_epoch": ep * 1e3, "pretrain_loss_first_last": [l0, l1],
           "proxy_dataset_frames_per_rank": n_frames, "proxy_dataset_ms_per_frame": pd / n_frames * 1e3,
           "proxy_dataset_mrays_per_s": world * n_frames * F_ * F_ / pd / 1e6,
           "proxy_truth_mrays_per_s": world * args.num_rays / proxy / 1e6, "proxy_truth_ms_per_batch": proxy * 1e3,
           "proxy_truth_note": "SealSteps.proxy_truth on one ray batch, teacher in eval mode (inference loop, eager launches)",
           "seal_train_samples_per_s": n_samples / reps["step"] / step, "seal_train_ms_per_step": step * 1e3,
           "seal_train_ms_per_step_with_collate": step_collate * 1e3,
           "seal_train_ms_per_step_online_proxy": step_online * 1e3,
           "seal_train_samples_per_s_online_proxy": n_on / reps["online"] / step_online,
           "seal_train_ms_per_step_one_march": step_one_march * 1e3,
           "seal_train_samples_per_s_one_march": n_om / reps["step"] / step_one_march,
           "proxy_truth_ms_per_batch_graph_replay": None if proxy_graph is None else proxy_graph * 1e3,
           "proxy_truth_mrays_per_s_graph_replay": None if proxy_graph is None else world * args.num_rays / proxy_graph / 1e6,
           "graph_captures": getattr(tr, "n_captures", 0), "lr_after": [float(g_["lr"]) for g_ in tr.optimizer.param_groups][:1]}
    if dps is not None:
        # the step's one exchange, alone: both tables' + both MLPs' gradients (fp16 hand-over buffer + fp32 bucket)
        ar = sync_time(lambda i: dps.allreduce_grads(tr.scaler), reps["allreduce"])
        half = sum(b.numel() * b.element_size() for b in dps.half_grads)
        flat = 0 if dps.flat is None else dps.flat.numel() * dps.flat.element_size()
        seen = torch.ones(1, device=dev)
        if multi:
            dist.all_reduce(seen)
        out["data_parallel"] = {
            "ranks_seen": int(seen.item()), "backend": dist.get_backend() if dist.is_initialized() else None,
            "allreduce_ms_per_step_alone": ar * 1e3, "allreduce_bytes_fp16_buffer": int(half), "allreduce_bytes_fp32_bucket": int(flat),
            "allreduce_in_step_graph": bool(getattr(tr, "collectives_in_graph", False)),
            "pretraining": f"each {'chunk'} of local points sharded over {world} rank(s), table gradients all-reduced, steps eager" if world > 1
                           else "1 rank: the chunk's step replayed from its graph",
            "finetune": f"{args.num_rays} rays per rank and step (weak scaling), proxy targets rendered by each rank for its own rays"}
    return out


# ----------------------------------------------------------------------------- configs[4]: TensoRF VM-48
def tensorf_section(args, dev, batches, note=lambda m: None, res=300, steps=24):
    """BASELINE configs[4]: the TensoRF VM-48 backbone (tensoRF/network.py: density rank 16x3, colour rank 48x3, basis 144 -> 27,
    colour MLP 150 -> 128 -> 128 -> 3) at resolution 300 on the synthetic scene, the reference's training step (tensoRF/utils.py:
    NGP step + L1 penalty on the density factors, weight 1e-4; lr 2e-2 factors / 1e-3 networks) — NativeAdam, fused VM feature
    kernels (csrc/tensorf.hip), timed eagerly and replayed from a HIP graph (`ms_per_step`).  `roofline`: the colour factors' backward (s3d_vm_color_backward: bound + plane + line
    kernels and the zero fills of its buffers, HIP events on the launch stream) against its algorithmic bytes per sample:
    3 components x (4 corners x 48 ranks x 4 B read + the same added, 2 x 48 x 4 B line values, 48 x 4 B g m written and read,
    2 x 48 x 4 B line gradient) + 12 B position + 64 B output gradient = 8,140 B."""
    import s3d_hip
    from nerf import synthetic as syn
    from tensoRF import network as trf
    from tensoRF.utils import GraphedTrainer as TensoRFGraphedTrainer, Trainer as TensoRFTrainer
    torch.manual_seed(args.seed + 31)
    net = trf.NeRFNetwork(resolution=[res] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
    grid, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(grid))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    net.iter_density = 100
    tr = TensoRFTrainer(net, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True, update_extra_interval=10 ** 9)
    tr.global_step = 1
    for k in range(4):
        tr.train_step(*batches[k % len(batches)])
    net.mean_count = int(net.step_counter[:4, 0].float().mean().item())  # the sample budget after the first grid update
    net.local_step = 0
    for k in range(4):
        tr.train_step(*batches[k % len(batches)])
    timers = KernelTimers(s3d_hip.VmBackend, ["color_backward", "features_backward"])
    # (a ~0.25 ms GPU spin in front of each bracket: the call is three zero fills, two allocations and seven launches — without the
    #  head start the bracket counts the host's launch latency between them, 426 us against ~340 us of kernels in the trace)
    timers.install(lambda name, a: a[0].shape[0], queue_ahead=600000)
    for k in range(4):
        tr.train_step(*batches[k % len(batches)])
    torch.cuda.synchronize()
    op = timers.summary()
    timers.remove()

    def timed(trainer, n_steps):
        samples = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_steps):
            trainer.train_step(*batches[k % len(batches)])
            samples.add_(net.step_counter[(net.local_step - 1) % 16, 0].long())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_steps, float(samples.item()) / n_steps

    for k in range(2):
        tr.train_step(*batches[k % len(batches)])
    dt_eager, n_eager = timed(tr, max(steps // 2, 4))
    # the same step replayed from a HIP graph (tensoRF/utils.py: GraphedTrainer) — the eager step is bound by its ~150 launches
    gtr = TensoRFGraphedTrainer(net, args.num_rays, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True, update_extra_interval=10 ** 9)
    gtr.global_step = 1
    for k in range(6):
        gtr.train_step(*batches[k % len(batches)])
    dt_graph, n_graph = timed(gtr, steps)
    # (both are product routes: the replayed step carries a static sample budget 1.3x the marched count, and the kernels without a
    #  device-side row count — the W = 128 MLP — process the padding; since the step is down to ~60 launches the eager one, with the
    #  exact budget, can be the faster of the two)
    dt = min(dt_graph, dt_eager)
    n = n_graph if dt_graph <= dt_eager else n_eager  # (the sample count of the SAME run the reported time comes from)
    out = {"workload": f"configs[4]: TensoRF VM-48 (sigma rank 16x3, colour rank 48x3), resolution {res}, {args.num_rays} rays/step, "
                       "training step with the L1 penalty (weight 1e-4), NativeAdam, fused VM kernels; the faster of eager and HIP-graph replay",
           "ms_per_step": dt * 1e3, "trainer": "graph" if dt_graph <= dt_eager else "eager", "ms_per_step_graph": dt_graph * 1e3,
           "ms_per_step_eager": dt_eager * 1e3, "graph_captures": gtr.n_captures, "samples_per_step": n, "samples_per_s": n / dt,
           "samples_per_s_graph": n_graph / dt_graph, "samples_per_s_eager": n_eager / dt_eager}
    cb = op.get("color_backward")
    if cb:
        bytes_per = 3 * (2 * 4 * 48 * 4 + 2 * 48 * 4 + 2 * 48 * 4 + 2 * 48 * 4) + 12 + 64
        ach = bytes_per * cb["units"] / (cb["avg_us"] * 1e-6) / 1e9
        out["roofline"] = {"kernel": "s3d_vm_color_backward (bound + plane + line + flush-reduce kernels, with the call's zero fills)", "bound": "hbm",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "algorithmic_bytes_per_sample": bytes_per, "avg_us": cb["avg_us"], "rows": cb["units"], "traffic": None}
        # counter traffic of the same call (FETCH_SIZE x 2 + WRITE_SIZE of its kernels, tools/pmc_tensorf_traffic.sh), quoted only while
        # tensorf.hip still has the digest the passes ran on
        import glob as _glob
        pm = sorted(_glob.glob(os.path.join(REPO, "profiles", "r*_tensorf_pmc.json")))
        if pm:
            try:
                rec = json.load(open(pm[-1]))
                if rec.get("tensorf_hip_sha256_16") == source_digest(os.path.join("seal-3d_amd", "csrc", "tensorf.hip")):
                    out["roofline"]["traffic"] = rec["color_backward_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(pm[-1])
                else:
                    out["roofline"]["traffic_source"] = f"profiles/{os.path.basename(pm[-1])} is stale (measured on another tensorf.hip)"
            except (OSError, ValueError, KeyError):
                pass
    fb = op.get("features_backward")
    if fb:
        out["density_factor_backward_us"] = fb["avg_us"]
    return out


def seal_tensorf_section(args, dev, batches, note=lambda m: None, res=300):
    """BASELINE configs[4] as `main_SealTensoRF.py:14-17` runs it: a TensoRF VM-48 teacher behind the bbox proxy and a TensoRF student
    (`sealnerf.get_trainer("tensorf")`: Seal's steps on the TensoRF trainer — nothing frozen during local pretraining, the L1 penalty
    inside every fine-tuning step), same edit and lattice step as the NGP Seal section; the student's step is replayed from a HIP graph,
    the teacher's proxy render is launched eagerly (its kernels take no device-side row count)."""
    from nerf import synthetic as syn
    from sealnerf import SealBBoxMapper, get_trainer, make_student, make_teacher
    from tensoRF import network as trf
    from tensoRF.utils import Trainer as TensoRFTrainer
    torch.manual_seed(args.seed + 47)
    kw = dict(resolution=[res] * 3, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    teacher = make_teacher(trf.NeRFNetwork, **kw).to(dev)
    grid, bits = syn.lego_like_density_grid(seed=0)
    teacher.density_grid.copy_(torch.from_numpy(grid))
    teacher.density_bitfield.copy_(torch.from_numpy(bits))
    teacher.iter_density = 100
    ttr = TensoRFTrainer(teacher, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True, update_extra_interval=10 ** 9)
    ttr.global_step = 1
    for i in range(48):  # a teacher that shows the scene: a few dozen steps on the training rays
        ttr.train_step(*batches[i % len(batches)])
    del ttr
    student = make_student(trf.NeRFNetwork, **kw).to(dev)
    student.load_state_dict(teacher.state_dict())
    student.mean_count, student.mean_density, student.iter_density = teacher.mean_count, teacher.mean_density, teacher.iter_density
    mapper = SealBBoxMapper(SEAL_BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    tr = get_trainer("tensorf", graphed=True)(student, teacher, args.num_rays, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True,
                                              update_extra_interval=10 ** 9)
    n_local = tr.init_pretraining(batch_size=6144000, lr=0.02, local_point_step=args.seal_point_step)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    l0 = float(tr.pretrain_one_epoch())
    tr.pretrain_one_epoch()
    ep = timed(lambda i: tr.pretrain_one_epoch(), 4)
    l1 = float(tr.pretrain_one_epoch())
    note(f"seal tensorf: pretraining timed ({ep * 1e3:.2f} ms/epoch)")
    tr.global_step = 1
    teacher.train(False)  # main_SealTensoRF.py:184
    proxy = timed(lambda i: tr.proxy_truth(batches[i % len(batches)][0], batches[i % len(batches)][1]), 4)
    nb = min(8, len(batches))
    # the targets the proxied dataset would hold for these rays (SealNeRF/provider.py:19-70: rendered once, before training)
    targets = [tuple(t.reshape(-1, t.shape[-1]) if t.ndim == 3 else t.reshape(-1) for t in tr.proxy_truth(batches[i][0], batches[i][1]))
               for i in range(nb)]
    for i in range(8):
        tr.train_step(batches[i % nb][0], batches[i % nb][1], *targets[i % nb])
    student.mean_count = int(student.step_counter[:8, 0].float().mean().item())  # the sample budget after the first grid update
    student.local_step = 0
    for i in range(16):  # (capture of the student's step, replays)
        tr.train_step(batches[i % nb][0], batches[i % nb][1], *targets[i % nb])
    samples = torch.zeros(1, dtype=torch.int64, device=dev)

    def ft(i):
        tr.train_step(batches[i % nb][0], batches[i % nb][1], *targets[i % nb])
        samples.add_(student.step_counter[(student.local_step - 1) % 16, 0].long())
    step = timed(ft, 16)
    n = float(samples.item()) / 16
    step_online = timed(lambda i: tr.train_step(batches[i % len(batches)][0], batches[i % len(batches)][1]), 8)
    return {"workload": f"configs[4] (main_SealTensoRF.py): lego_bbox-shaped edit, TensoRF VM-48 teacher + student at resolution {res}, "
                        f"pretraining_local_point_step={args.seal_point_step:g}, {args.num_rays} rays/step, targets proxied before training, "
                        "student step replayed from a HIP graph",
            "graph_captures": getattr(tr, "n_captures", 0),
            "local_points": int(n_local), "pretrain_ms_per_epoch": ep * 1e3, "seal_pretrain_points_per_s": n_local / ep,
            "pretrain_loss_first_last": [l0, l1], "proxy_truth_ms_per_batch": proxy * 1e3,
            "seal_train_ms_per_step": step * 1e3, "seal_train_samples_per_s": n / step, "samples_per_step": n,
            "seal_train_ms_per_step_online_proxy": step_online * 1e3,
            "note": "a fine-tuning step = the student's training step on targets the eval-mode teacher rendered beforehand "
                    "(main_SealTensoRF.py's flow: proxy_dataset, then skip_proxy batches); *_online_proxy adds the GUI loop's "
                    "per-step teacher render (inference loop, eager)"}


# ----------------------------------------------------------------------------- quality over a long run
def long_run_quality(args, dev, R, scene_bits, boxes, steps=3000, note=lambda m: None, seeds=3, n_views=4, hw=200):
    """Does the native fp16 path (fp16 table gradients, exact fixed-point sums, native Adam + loss scaling, HIP-graph replay)
    CONVERGE like the reference arrangement (torch.optim.Adam on fp32 `.grad`s + torch GradScaler, eager)?  Both train
    configs[1]'s network from the same initial weights on the same batches for `steps` steps (lr 1e-2 decayed to 0.1x as
    main_SealNeRF.py:283-288, every step in both arrangements: the replayed graph reads the schedule's factor from a device
    word, nerf/optim.py: follow_lr_schedule); PSNR (nerf/utils.py:226-233) on four HELD-OUT 200x200 views against the analytic scene.
    The two trajectories are chaotic twins (different rounding, different RNG consumption under capture), so ONE pair says
    little: the comparison is repeated for `seeds` initialisations and reported as mean +- sample standard deviation of
    each arrangement and of the paired difference."""
    from nerf import network_ff, synthetic as syn
    from nerf.trainer import GraphedTrainer, Trainer, psnr
    kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10)
    pool, _ = make_batches(768, args.num_rays, 4242, dev, R, scene_bits, boxes)  # 3.1 M distinct rays of the 100 training cameras
    views = syn.orbit_poses(n_views, seed=977)  # not among the 100 training cameras (seed 0)
    rays = [syn.get_rays(views[i:i + 1].to(dev), syn.lego_intrinsics(hw, hw), hw, hw) for i in range(n_views)]
    gts = [analytic_targets(r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous(), scene_bits, boxes, R) for r in rays]
    tags = (("native_fp16_graph", True), ("torch_adam_fp32_eager", False))
    runs = {t: [] for t, _ in tags}
    for k in range(seeds):
        torch.manual_seed(args.seed + 5 + 101 * k)
        init = network_ff.NeRFNetwork(**kw).to(dev).state_dict()
        for tag, native in tags:
            torch.manual_seed(args.seed + 6 + 101 * k)
            m = network_ff.NeRFNetwork(**kw).to(dev)
            m.load_state_dict(init)
            tr = GraphedTrainer(m, args.num_rays, lr=1e-2, fp16=True) if native else Trainer(m, lr=1e-2, fp16=True, native_optim=False)
            t0 = time.perf_counter()
            for i in range(steps):
                lr = 1e-2 * 0.1 ** min(i / steps, 1.0)
                for g in tr.optimizer.param_groups:
                    g["lr"] = lr
                tr.train_step(*pool[(i + 257 * k) % len(pool)])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            vals = []
            for r, gt in zip(rays, gts):
                img = tr.render_image(r["rays_o"].contiguous(), r["rays_d"].contiguous())["image"][0]
                vals.append(psnr(img, gt))
            runs[tag].append({"psnr_db_per_view": [round(v, 3) for v in vals], "psnr_db": float(np.mean(vals)), "train_s": round(dt, 2)})
            note(f"long run [{tag}, seed {k}]: {np.mean(vals):.2f} dB in {dt:.1f} s")
            del tr, m
    out = {}
    for tag, _ in tags:
        v = [r["psnr_db"] for r in runs[tag]]
        out[tag] = {"psnr_db": float(np.mean(v)), "psnr_db_std": float(np.std(v, ddof=1)) if len(v) > 1 else 0.0,
                    "psnr_db_per_seed": [round(x, 3) for x in v], "runs": runs[tag]}
    d = [a["psnr_db"] - b["psnr_db"] for a, b in zip(runs["native_fp16_graph"], runs["torch_adam_fp32_eager"])]
    out["steps"], out["seeds"] = steps, seeds
    out["delta_db_per_seed"] = [round(x, 3) for x in d]
    out["delta_db"] = float(np.mean(d))
    out["delta_db_std"] = float(np.std(d, ddof=1)) if len(d) > 1 else 0.0
    # the paired difference's standard error and 95 % interval (Student t): per-seed differences scatter with sigma ~0.25 dB at
    # 6,000 steps / 16 views of 400x400 (profiles/r11_psnr_seeds96.json; ~0.5 dB at 3,000 steps / 4 views of 200x200,
    # profiles/r10_psnr_seeds*.json), so a mean of few seeds cannot resolve 0.1 dB: the 96-seed study settles it
    sem = out["delta_db_std"] / math.sqrt(len(d)) if len(d) > 1 else float("nan")
    t975 = {2: 12.706, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365, 12: 2.201, 16: 2.131, 32: 2.040, 64: 1.998}
    tq = t975[max(k for k in t975 if k <= max(len(d), 2))]
    out["delta_db_sem"] = sem
    out["delta_db_ci95"] = [out["delta_db"] - tq * sem, out["delta_db"] + tq * sem]
    out["within_0p1_db"] = bool(abs(out["delta_db"]) <= 0.1)
    out["ci95_overlaps_0p1_db"] = bool(out["delta_db_ci95"][0] <= 0.1 and out["delta_db_ci95"][1] >= -0.1)
    out["views"] = f"{n_views} held-out {hw}x{hw} orbit cameras (seed 977), analytic box scene"
    return out


# ----------------------------------------------------------------------------- main
def main():
    args = parse()
    from parallel import RayShardedDP, init_from_env, launched_by_torchrun, spawn_ranks
    if args.gpus > 1 and not launched_by_torchrun():
        # `python bench.py --gpus N` typed as it stands: become N ranks (one per GPU) under torch.distributed.run
        spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], need_gpus=args.backend != "gloo")
    rank, world, local = init_from_env(args.backend)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus}, but {world} rank(s) were launched (WORLD_SIZE): refusing to report a number "
                 "for another GPU count than the one asked for")
    if args.rendezvous_only:
        import torch.distributed as dist
        seen = torch.ones(1, device=torch.device("cuda", local) if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(seen)
        if world > 1:
            dist.destroy_process_group()
        if int(seen.item()) != args.gpus:
            sys.exit(f"bench.py: {int(seen.item())} ranks answered the all-reduce, {args.gpus} expected")
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": int(seen.item()), "backend": args.backend or "nccl"}), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} GPUs requested, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import s3d_hip
    s3d_hip.lib()
    from nerf import network, network_ff, synthetic as syn
    from nerf.trainer import GraphedTrainer, Trainer, psnr
    import torch.distributed as dist

    torch.manual_seed(args.seed)
    Net = network_ff.NeRFNetwork if args.net == "ff" else network.NeRFNetwork
    model = Net(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).to(dev)
    if args.force_dp and world == 1:  # single-GPU exercise of the data-parallel code path (1-rank RCCL group)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dp = RayShardedDP(force_collective=args.force_dp) if (world > 1 or args.force_dp) else None
    if args.no_graph:
        trainer = Trainer(model, lr=1e-2, fp16=True, update_extra_interval=16, dist=dp)
    else:
        trainer = GraphedTrainer(model, args.num_rays, lr=1e-2, fp16=True, update_extra_interval=16, dist=dp)

    R = s3d_hip.RaymarchingBackend
    grid, bits = syn.lego_like_density_grid(seed=0)
    scene_bits = torch.from_numpy(bits).to(dev)
    boxes = syn.lego_like_boxes(0)
    n_pool = 32
    batches, poses = make_batches(n_pool, args.num_rays, args.seed + rank, dev, R, scene_bits, boxes)

    def step(i):
        ro, rd, gt = batches[i % n_pool]
        return trainer.train_step(ro, rd, gt)  # (with dp: occupancy state is synchronised inside, after each grid update)

    # --- setup: converge the occupancy grid (untimed, not part of warm-up)
    for i in range(args.pretrain):
        step(i)
    # --- warm-up
    for i in range(args.warmup):
        step(i)

    timers = KernelTimers(s3d_hip.GridBackend, ["grid_encode_forward", "grid_encode_backward", "grid_encode_backward_adam"])
    rt = KernelTimers(s3d_hip.RaymarchingBackend, ["march_rays_train", "composite_rays_train_forward", "composite_rays_train_backward"])
    ft = KernelTimers(s3d_hip.FFMLPBackend, ["ffmlp_forward", "ngp_pair_inference", "ffmlp_backward"])
    graphed = not args.no_graph

    def install_timers(queue_ahead=False):
        timers.install(grid_meta, queue_ahead)
        rt.install(lambda n, a: 0, queue_ahead)
        ft.install(lambda n, a: a[2] if n == "ffmlp_forward" else (a[3] if n == "ngp_pair_inference" else a[4]), queue_ahead)
    if not graphed:
        install_timers()

    # density-grid maintenance (every 16 steps, inside the timed region like in the reference's train loop): event pair per call
    ues_events = []
    ues_inner = trainer._maybe_update_extra_state

    def timed_update_extra_state(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = ues_inner(*a, **k)
        e1.record()
        if r:  # (False: not an update step)
            ues_events.append((e0, e1))
        return r
    trainer._maybe_update_extra_state = timed_update_extra_state

    samples_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    captures0 = getattr(trainer, "n_captures", 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # samples are counted from the renderer's own 16-step counter ring, read when it is about to be restarted (not per step:
    # the bookkeeping is not part of the reference's step either)
    ring_from = model.local_step % 16
    for i in range(args.steps):
        step(i)
        if model.local_step % 16 == 0 or trainer.global_step % trainer.update_extra_interval == 0:
            samples_dev += model.step_counter[ring_from:(model.local_step - 1) % 16 + 1, 0].sum()
            ring_from = 0 if trainer.global_step % trainer.update_extra_interval == 0 else model.local_step % 16
    if model.local_step % 16 != ring_from:
        samples_dev += model.step_counter[ring_from:model.local_step % 16, 0].sum()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    trainer._maybe_update_extra_state = ues_inner
    ues_ms = [a.elapsed_time(b) for a, b in ues_events]
    # secondary figure over (at least) 64 steps = four whole occupancy-update periods: `ms_per_step` of a short timed region moves
    # by +-5 % with where the 16-step boundary falls (one 0.5 ms update inside 20 steps is 5 % of them)
    long_steps = max(64, (args.steps + 15) // 16 * 16)
    s64 = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t64 = time.perf_counter()
    ring_from = model.local_step % 16
    for i in range(long_steps):
        step(args.steps + i)
        if model.local_step % 16 == 0 or trainer.global_step % trainer.update_extra_interval == 0:
            s64 += model.step_counter[ring_from:(model.local_step - 1) % 16 + 1, 0].sum()
            ring_from = 0 if trainer.global_step % trainer.update_extra_interval == 0 else model.local_step % 16
    if model.local_step % 16 != ring_from:
        s64 += model.step_counter[ring_from:model.local_step % 16, 0].sum()
    torch.cuda.synchronize()
    t64 = time.perf_counter() - t64
    long_run = {"steps": long_steps, "ms_per_step": t64 / long_steps * 1e3, "samples_per_s_this_rank": float(s64.item()) / t64}
    timer_steps = args.steps
    if graphed:
        # HIP events cannot be recorded inside a graph replay: time the individual kernels in an eager pass of the
        # SAME step on the same state, immediately after the timed region (not part of `value`)
        eager = Trainer(model, lr=1e-2, fp16=True, update_extra_interval=10 ** 9, dist=None, optimizer=trainer.optimizer,
                        scaler=trainer.scaler)
        eager.global_step = 1
        install_timers(queue_ahead=True)
        timer_steps = 32
        for i in range(-2, timer_steps):
            if i == 0:  # (the first step of either kind pays first-use allocations: not kept)
                torch.cuda.synchronize()
                for t in (timers, rt, ft):
                    t.reset()
            ro, rd, gt = batches[i % n_pool]
            # (half of the pass with the table's update as its own launch: the plain scatter + accumulate pair stays measured,
            #  `roofline_grid_backward_plain`, next to the product's in-backward update)
            eager.fuse_table_updates = trainer.fuse_table_updates and i % 2 == 0
            eager.train_step(ro, rd, gt)
        torch.cuda.synchronize()
    for t in (timers, rt, ft):
        t.remove()

    samples = float(samples_dev.item())
    if world > 1:
        tt = torch.tensor([elapsed, samples], dtype=torch.float64, device=dev)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed, samples = float(mx[0]), float(tt[1])

    # data-parallel runs: what the gradient all-reduce costs on its own on every rank (eager, outside the timed region), and how
    # many ranks actually took part — so that a scaling curve can be read against the wire time
    dp_info = None
    if dp is not None:
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        for _ in range(3):
            dp.allreduce_grads()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(10):
            dp.allreduce_grads()
        a1.record()
        torch.cuda.synchronize()
        mine = torch.tensor([a0.elapsed_time(a1) / 10], dtype=torch.float32, device=dev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        nbytes = sum(h.numel() * h.element_size() for h in dp.half_grads) + (dp.flat.numel() * 4 if dp.flat is not None else 0)
        dp_info = {"n_ranks_seen": int(seen.item()), "allreduce_ms_per_rank": [round(float(t.item()), 4) for t in per_rank],
                   "gradient_bytes": int(nbytes), "collectives_in_graph": bool(getattr(trainer, "collectives_in_graph", False)),
                   "reduce_op": "avg (in the collective)" if dp.fused_avg() else "sum + divide"}
        if dp_info["n_ranks_seen"] != args.gpus:
            sys.exit(f"bench.py: {dp_info['n_ranks_seen']} ranks took part in the gradient all-reduce, --gpus {args.gpus}")
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # --- roofline of the dominant hot kernel
    ksum = {}
    for t in (timers, rt, ft):
        ksum.update(t.summary())
    s = 2  # fp16 tables under -O
    bytes_pt = 12 + 16 * 8 * 2 * s + 16 * 2 * s  # 588 B / point / encoder
    # (per launch: a step has one of grid_encode_backward / grid_encode_backward_adam; the eager timing pass alternates them)
    dom = max(("grid_encode_forward", "grid_encode_backward", "grid_encode_backward_adam"), key=lambda n: ksum.get(n, {}).get("avg_us", 0))
    kd = ksum[dom]
    n_params = sum(p.numel() for p in model.parameters() if p.dim() == 2 and p.shape[1] == 2 and p.shape[0] > 100000)
    # the in-backward update moves the optimizer's 30 B per parameter (p, m, v read and written in fp32, fp16 copy written,
    # gradient written + read + cleared) as part of the op: VERDICT r5 #4's bookkeeping
    op_bytes = kd["units"] * bytes_pt + (30.0 * n_params if dom == "grid_encode_backward_adam" else 0.0)
    achieved = op_bytes / (kd["avg_us"] * 1e-6) / 1e9
    traffic, traffic_src = profile_traffic(dom)
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "avg_us": kd["avg_us"], "points_per_launch": kd["units"],
                "algorithmic_bytes_per_point": bytes_pt, "algorithmic_bytes_per_launch": op_bytes,
                "algorithmic_bytes_note": ("588 B x points + 30 B x table parameters (the Adam update rides in the accumulate kernel)"
                                           if dom == "grid_encode_backward_adam" else "588 B x points"),
                "kernels_ms_per_step": {k: v["total_ms"] / timer_steps for k, v in ksum.items()},
                "kernel_calls_in_timing_pass": {k: v["calls"] for k, v in ksum.items()},
                "timing": "HIP events around each native call on the launch stream; " +
                          ("eager pass of 32 identical steps (2 more discarded) right after the graph-replayed timed region, each call queued behind "
                           "a short GPU spin so that its kernels run back to back as in the graph" if graphed
                           else "inside the timed region")}

    if "grid_encode_backward" in ksum:  # the plain scatter + accumulate pair (gradient table written), as rounds 3 - 5 reported it
        kb = ksum["grid_encode_backward"]
        ach_b = kb["units"] * bytes_pt / (kb["avg_us"] * 1e-6) / 1e9
        roofline["grid_backward_plain"] = {"kernel": "grid_encode_backward (k_bin_scatter6 + k_bin_accumulate6, update separate)",
                                           "achieved": ach_b, "frac": ach_b / HBM_PEAK_GBS, "avg_us": kb["avg_us"],
                                           "points_per_launch": kb["units"], "algorithmic_bytes_per_point": bytes_pt}
    # --- matrix-core roofline of the fused MLPs (north_star: MFMA utilisation on ffmlp against chip peak; SURVEY §8d: the
    # bound is min(MFMA peak, arithmetic intensity x HBM peak)).  Both ffmlp nets of this config are 32-64-64-16 (3 matmuls,
    # 7,168 MAC per sample forward); the training forward stores no activations (the fused backward re-computes them), so the
    # forward moves 2*32 + 2*16 bytes per sample and the backward 2*32 (input) + 2*16 (output grad) + 2*32 (input grad).
    # Backward flops: re-computed forward + data gradient + weight gradient = 3x the forward's.
    mfma_peak = 2500.0  # TFLOP/s dense fp16, MI355X_MICROARCH.md
    roofline_ffmlp = {}
    def mlp_macs(net):  # MAC per sample of an FFMLP: in x W + (layers - 1) x W x W + W x 16 (padded output)
        return net.input_dim * net.hidden_dim + (net.num_layers - 1) * net.hidden_dim ** 2 + net.hidden_dim * 16
    nets = [n for n in (getattr(model, "sigma_net", None), getattr(model, "color_net", None)) if hasattr(n, "num_layers")]
    # (the timers average over the launches of both networks: density net 32-64-64-16 = 7,168 MAC, colour net 32-64-64-64-16 =
    #  11,264 MAC per sample; rounds 1-3 priced both at 7,168)
    macs = sum(mlp_macs(n) for n in nets) / len(nets) if nets else 7168
    # (round 4: the training forward of both networks is ONE launch, `ngp_pair_inference` = k_ffmlp_ngp_pair: both networks'
    #  flops; 64 B encoder features + 12 B directions in, 64 B colour-net input + 4 + 2 + 12 B out per sample)
    for name, flops, byts in (("ffmlp_forward", 2 * macs, 96), ("ngp_pair_inference", 2 * macs * len(nets), 158),
                              ("ffmlp_backward", 6 * macs, 160)):
        k = ksum.get(name)
        if not k or not k["units"]:
            continue
        ach = k["units"] * flops / (k["avg_us"] * 1e-6) / 1e12
        bound = min(mfma_peak, flops / byts * HBM_PEAK_GBS / 1e3)
        roofline_ffmlp[name] = {"achieved": ach, "unit": "TFLOP/s", "bound": bound, "bound_is": "mfma" if bound == mfma_peak else "AI x hbm",
                                "frac_of_bound": ach / bound, "frac_of_mfma_peak": ach / mfma_peak, "avg_us": k["avg_us"],
                                "samples_per_launch": k["units"], "flop_per_sample": flops, "algorithmic_bytes_per_sample": byts}
    if roofline_ffmlp:
        import glob as _glob
        _tr = sorted(f for f in _glob.glob(os.path.join(REPO, "profiles", "r*_timed_region.md")) if re.match(r"r\d+_timed_region\.md$", os.path.basename(f)))
        roofline_ffmlp["counters"] = ((f"profiles/{os.path.basename(_tr[-1])}" if _tr else "no committed profile") +
                                      ", section 'Matrix cores' (SQ_INSTS_VALU_MFMA_MOPS_F16, SQ_VALU_MFMA_BUSY_CYCLES per kernel); issued MFMA "
                                      "work = 1.14x / 1.09x the algorithmic work of the density / colour net (16-row output layer on a 32-row tile)")

    extra = {"roofline_ffmlp": roofline_ffmlp, "samples_per_s_64steps": long_run}
    collectives_in_graph = bool(getattr(trainer, "collectives_in_graph", False))  # (the Seal section below drops `trainer`)
    if graphed:
        extra["graph_captures_in_timed_region"] = trainer.n_captures - captures0
    if ues_ms:
        extra["update_extra_state"] = {"calls_in_timed_region": len(ues_ms), "ms_per_call": sum(ues_ms) / len(ues_ms),
                                       "ms_per_step_amortised": sum(ues_ms) / args.steps}
    if args.save_model and rank == 0:
        torch.save(model.state_dict(), args.save_model)
    if not args.no_render:
        # full 800x800 frame renders (inference loop) + PSNR against the analytic scene
        model.infer_batch_scale = args.infer_batch_scale
        r = syn.get_rays(poses[:1].to(dev), syn.lego_intrinsics(), 800, 800)
        ro, rd = r["rays_o"].contiguous(), r["rays_d"].contiguous()
        out = trainer.render_image(ro, rd)
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        nfr = 3
        for _ in range(nfr):
            out = trainer.render_image(ro, rd)
        torch.cuda.synchronize()
        dtr = (time.perf_counter() - tr0) / nfr
        gt = torch.cat([analytic_targets(ro[0, i:i + 160000].contiguous(), rd[0, i:i + 160000].contiguous(), scene_bits, boxes, R)
                        for i in range(0, 640000, 160000)])
        extra.update({"render_infer_batch_scale": args.infer_batch_scale, "render_mrays_per_s": 0.64 / dtr,
                      "render_ms_per_frame": dtr * 1e3, "psnr_vs_analytic_scene": psnr(out["image"][0], gt)})
        # roofline of the render's dominant kernel (the hash-grid forward of the inference loop): one more frame with HIP events
        # around every grid_encode_forward call; points = the rows the kernel actually works on (device-side counts, n_valid)
        rtm = KernelTimers(s3d_hip.GridBackend, ["grid_encode_forward"])
        rtm.install(grid_meta)
        trainer.render_image(ro, rd)
        torch.cuda.synchronize()
        rtm.remove()
        rk = rtm.summary().get("grid_encode_forward")
        if rk:
            pts = rk["units"] * rk["calls"]
            ach = pts * 588 / (rk["total_ms"] * 1e-3) / 1e9
            extra["roofline_render"] = {"kernel": "grid_encode_forward (inference loop)", "bound": "hbm", "achieved": ach,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                        "points_per_frame": pts, "launches_per_frame": rk["calls"],
                                        "grid_forward_ms_per_frame": rk["total_ms"], "frame_ms": dtr * 1e3,
                                        "mrays_per_s": 0.64 / dtr, "algorithmic_bytes_per_point": 588}
            try:  # every kernel launch of one frame (torch's profiler counts device kernels, native and torch alike)
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    trainer.render_image(ro, rd)
                    torch.cuda.synchronize()
                extra["roofline_render"]["kernel_launches_per_frame"] = sum(
                    1 for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
            except Exception as e:  # (profiler unavailable: the figure is optional)
                extra["roofline_render"]["kernel_launches_per_frame"] = None
                extra["roofline_render"]["kernel_launches_note"] = f"torch.profiler failed: {type(e).__name__}"

    def note(msg):
        print(f"[bench] {msg}", file=sys.stderr, flush=True)
    if world == 1 and not args.no_render:
        note("psnr vs oracle render")
        extra["psnr"] = psnr_vs_oracle(model, lambda: Net(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10),
                                       poses, scene_bits, boxes, dev, R)
    if world == 1 and not args.no_long_run and args.net == "ff":
        lr_ = long_run_quality(args, dev, R, scene_bits, boxes, steps=args.long_run_steps, note=note, seeds=args.long_run_seeds,
                               n_views=args.long_run_views, hw=400)
        # the committed many-seed study of the same comparison (tools/psnr_seeds.py): what settles "within 0.1 dB"
        study = os.path.join(REPO, "profiles", "r11_psnr_seeds96.json")
        if os.path.exists(study):
            try:
                st = json.load(open(study))
                ci = st["delta_db_ci95"]
                lr_["many_seed_study"] = {"file": "profiles/r11_psnr_seeds96.json", "seeds": st["seeds"], "steps": st["steps"], "views": st.get("views"),
                                          "delta_db": st["delta_db"], "delta_db_sem": st["delta_db_sem"], "delta_db_std": st["delta_db_std"],
                                          "delta_db_ci95": ci, "ci95_half_width_db": (ci[1] - ci[0]) / 2,
                                          "within_0p1_db": st["within_0p1_db"], "ci95_inside_0p1_db": bool(ci[0] >= -0.1 and ci[1] <= 0.1)}
            except (OSError, ValueError, KeyError):
                pass
        extra.setdefault("psnr", {})["long_run"] = lr_
        ms = lr_.get("many_seed_study")
        if ms:  # north_star "PSNR within 0.1 dB of the reference path": settled by the committed study, the run above is a 16-seed sample of it
            extra["psnr"]["vs_reference_path"] = {"delta_db": ms["delta_db"], "ci95": ms["delta_db_ci95"], "seeds": ms["seeds"],
                                                  "within_0p1_db_with_95pct_confidence": ms["ci95_inside_0p1_db"], "source": ms["file"],
                                                  "this_run": {"seeds": lr_["seeds"], "delta_db": lr_["delta_db"], "ci95": lr_["delta_db_ci95"]}}
    if not args.no_seal and args.net == "ff":
        # configs[2] on one GPU; under `--gpus N` (or --force_dp) configs[3]: the same section data-parallel over the ranks
        note("seal section")
        del trainer
        try:
            extra["seal"] = seal_section(args, dev, batches, note if rank == 0 else (lambda m: None),
                                         make_dp=(lambda: RayShardedDP(force_collective=args.force_dp)) if dp is not None else None)
        except Exception as e:  # noqa: BLE001
            if world == 1:
                raise
            # the headline line of an N-rank run must not be lost to a failure of this optional section (every rank runs the
            # same code on the same shapes: a Python-level failure is raised on all of them)
            extra["seal"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if world == 1 and not args.no_tensorf and args.net == "ff":
        note("tensorf section")
        extra["tensorf"] = tensorf_section(args, dev, batches, note)
        if not args.no_seal:
            try:
                extra["tensorf"]["seal"] = seal_tensorf_section(args, dev, batches, note)
            except Exception as e:  # (a reported gap, not a lost bench line)
                extra["tensorf"]["seal"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        note("cpu baseline")
        cpu = cpu_baseline(args, args.cpu_rays)

    line = {
        "metric": "train samples/s (NGP -O step on synthetic Lego 800x800 rays)", "value": samples / elapsed, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "configs[1]: nerf_synthetic/lego-shaped NGP -O (hashgrid L16 F2 T2^19 + ffmlp 64x2/64x3 + raymarching), "
                               "800x800 cameras, 4096 rays/step/GPU" if args.net == "ff" else
                               "Seal NGP net (two hash encoders + nn.Linear MLPs), 800x800 cameras, 4096 rays/step/GPU",
                   "num_rays_per_gpu": args.num_rays, "samples_per_step": samples / args.steps / world,
                   "parallelism": f"ray-sharded dp{world}", "pretrain_steps": args.pretrain,
                   "launch": ("hip-graph replay" + ((" (one graph incl. the RCCL all-reduce)" if collectives_in_graph
                                                  else " (fwd+bwd | all-reduce | optimizer)") if dp is not None else "")) if graphed else "eager"},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    line.update(extra)
    # the second half of BASELINE.json's metric ("render Mrays/s"), also inside `config` (a field the driver parses)
    for k in ("render_mrays_per_s", "render_ms_per_frame"):
        if k in extra:
            line["config"][k] = extra[k]
            if "roofline_render" in extra:
                line["roofline_render"][k] = extra[k]
    # configs[2] / configs[4]: the figures of the optional sections, where the driver's parser looks (`config`)
    seal_, tf_ = extra.get("seal") or {}, extra.get("tensorf") or {}
    for key, src, name in (("seal_train_ms_per_step", seal_, "seal_train_ms_per_step"),
                           ("seal_pretrain_ms_per_epoch", seal_, "pretrain_ms_per_epoch"),
                           ("seal_proxy_dataset_mrays_per_s", seal_, "proxy_dataset_mrays_per_s"),
                           ("tensorf_ms_per_step", tf_, "ms_per_step"),
                           ("seal_tensorf_ms_per_step", tf_.get("seal") or {}, "seal_train_ms_per_step"),
                           ("seal_tensorf_pretrain_ms_per_epoch", tf_.get("seal") or {}, "pretrain_ms_per_epoch")):
        if isinstance(src.get(name), (int, float)):
            line["config"][key] = src[name]
    if dp_info is not None:
        line["data_parallel"] = dp_info
    if dist.is_initialized():
        dist.destroy_process_group()
    # RCCL writes a version banner to C stdout, which is flushed at exit — after Python's buffer.  Flush it now so the
    # JSON line is the LAST line of stdout.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
