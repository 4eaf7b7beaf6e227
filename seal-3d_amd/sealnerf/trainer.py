"""Teacher -> student distillation steps of Seal-3D (SealNeRF/trainer.py), data-parallel over points / rays.

* local pretraining (`init_pretraining` :88-157, `pretrain_step` :455-469): a dense lattice of points inside the edit
  bounds is mapped to source space, the teacher is queried ONCE for (sigma, colour) targets, then the student is fitted
  with L1(sigma) + L1(colour) on point batches, MLPs frozen (`freeze_mlp` :472-488) — pure encoder fwd/bwd, no marching;
* global fine-tuning (`train_step` :589-594 + nerf/utils.py:436-537): ordinary ray batches whose targets (RGB + depth)
  are rendered by the teacher through the proxy (`proxy_truth` :506-586); loss MSE(rgb) + L1(depth).
Sharding: every rank takes a contiguous shard of each point chunk / its own ray batch; gradients are summed
through `parallel.RayShardedDP`'s flat bucket (one RCCL all-reduce per step)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from nerf.trainer import _CAPTURE_MODE, GraphedTrainer, Trainer


def _euler_dirs(angle_step):
    """unit directions of `Rotation.from_euler('xyz', eulers, degrees=True).apply([1-1e-5, 0, 0])` (trainer.py:627-631)"""
    a = np.deg2rad(np.arange(0, 360, angle_step, dtype=np.float64))
    rx, ry, rz = np.meshgrid(a, a, a, indexing="ij")
    rx, ry, rz = rx.reshape(-1), ry.reshape(-1), rz.reshape(-1)
    # extrinsic xyz: R = Rz @ Ry @ Rx ; R @ [s,0,0] = s * first column
    s = 1 - 1e-5
    cy, sy, cz, sz = np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    return torch.from_numpy(np.stack([s * cz * cy, s * sz * cy, -s * sy], -1))


def sample_points(bounds, point_step=0.005, angle_step=45):
    """lattice points inside `bounds` ((2,3) or (B,2,3)) + the Euler-grid direction set — trainer.py:609-635"""
    if bounds.ndim == 2:
        bounds = bounds[None]
    pts, dirs = [], []
    for i in range(bounds.shape[0]):
        lo, hi = bounds[i].cpu()
        X, Y, Z = torch.meshgrid(torch.arange(lo[0], hi[0], step=point_step), torch.arange(lo[1], hi[1], step=point_step),
                                 torch.arange(lo[2], hi[2], step=point_step), indexing="ij")
        pts.append(torch.stack([X, Y, Z], dim=-1).reshape(-1, 3))
        dirs.append(_euler_dirs(angle_step))
    return torch.cat(pts), torch.cat(dirs)


class _L1Pair(torch.autograd.Function):
    """loss = L1Loss(sigma, gt_sigma) + L1Loss(color, gt_color) (means; SealNeRF/trainer.py:455-469) and, for the upstream gradient
    announced in advance (the loss scale), both gradients — one launch (csrc/ngp_head.hip: k_l1_pair) for sub / abs / sum / div x 2,
    add and their backward nodes.  `sigma` / `color` may carry padding rows behind the targets' n (no term, zero gradient)."""

    @staticmethod
    def forward(ctx, sigma, color, gt_sigma, gt_color, n_total, expected_grad=None):
        import s3d_hip
        sigma, color = sigma.float().contiguous(), color.float().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=sigma.device)
        ctx.pre = None
        if expected_grad is not None and expected_grad.dtype == torch.float32 and expected_grad.numel() == 1:
            g_sigma, g_color = torch.empty_like(sigma), torch.empty_like(color)
            s3d_hip.NgpHeadBackend.l1_pair_loss(sigma, color, gt_sigma, gt_color, n_total, loss, expected_grad, g_sigma, g_color)
            ctx.pre = (g_sigma, g_color, expected_grad.data_ptr(), expected_grad._version)
        else:
            s3d_hip.NgpHeadBackend.l1_pair_loss(sigma, color, gt_sigma, gt_color, n_total, loss)
        ctx.save_for_backward(sigma, color, gt_sigma, gt_color)
        ctx.n_total = n_total
        return loss

    @staticmethod
    def backward(ctx, g):
        import s3d_hip
        if ctx.pre is not None and g.dtype == torch.float32 and g.data_ptr() == ctx.pre[2] and g._version == ctx.pre[3]:
            return ctx.pre[0], ctx.pre[1], None, None, None, None
        sigma, color, gt_sigma, gt_color = ctx.saved_tensors
        g_sigma, g_color = torch.empty_like(sigma), torch.empty_like(color)
        scratch = torch.empty((), dtype=torch.float32, device=sigma.device)
        s3d_hip.NgpHeadBackend.l1_pair_loss(sigma, color, gt_sigma, gt_color, ctx.n_total, scratch, g.float().reshape(1).contiguous(),
                                            g_sigma, g_color)
        return g_sigma, g_color, None, None, None, None


def freeze_module(module, freeze):
    module.training = not freeze
    for p in module.parameters():
        p.requires_grad = not freeze


class SealSteps:
    """the distillation steps, shared by the eager and the graph-replayed trainer (mixed into a nerf.trainer.Trainer)"""

    def _init_seal(self, teacher, lr, depth_weight):
        self.teacher = teacher
        self.depth_weight = depth_weight
        self.pretraining_data = {}
        self.base_lr = lr

    # ------------------------------------------------------------------ local pretraining
    @torch.no_grad()
    def init_pretraining(self, batch_size=6144000, lr=0.05, local_point_step=0.005, local_angle_step=45, seed=0):
        mapper = self.teacher.seal_mapper
        dev = next(self.model.parameters()).device
        pts, dirs = sample_points(mapper.map_data["force_fill_bound"], local_point_step, local_angle_step)
        pts, dirs = pts.to(dev, torch.float32), dirs.to(dev, torch.float32)
        ones = torch.zeros_like(pts) + torch.tensor([1.0, 0, 0], device=dev)
        mapped_p, mapped_d, mask = mapper.map_to_origin(pts, ones)
        if "map_source" in mapper.map_data:
            mask[:] = True
        pts = pts[mask]
        g = torch.Generator(device="cpu").manual_seed(seed)
        dirs = dirs[torch.randint(dirs.shape[0], (pts.shape[0],), generator=g).to(dev)]
        mapped_p, mapped_d = mapped_p[mask], mapped_d[mask]
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16 and dev.type == "cuda"):
            chunks = [self.teacher(mapped_p[i:i + (1 << 20)], mapped_d[i:i + (1 << 20)]) for i in range(0, pts.shape[0], 1 << 20)]
        gt_sigma = torch.cat([c[0].float() for c in chunks])
        gt_color = mapper.map_color(mapped_p, mapped_d, torch.cat([c[1].float() for c in chunks]))
        steps = list(range(0, pts.shape[0], batch_size))
        if steps[-1] != pts.shape[0]:
            steps.append(pts.shape[0])
        self.pretraining_data["local"] = {"points": pts, "dirs": dirs, "sigma": gt_sigma, "color": gt_color, "steps": steps}
        self.pretraining_data.pop("_padded", None)
        self.pretraining_lr = lr
        self.invalidate_graphs()  # (the per-chunk graphs hold raw pointers into the previous point / target tensors)
        return pts.shape[0]

    def invalidate_graphs(self):
        """Drop every captured graph that bakes in state this trainer can replace: the per-chunk pretraining graphs (slices
        of `pretraining_data`, the optimizer's moment tensors, the learning rate), and the teacher's proxy-render graph (the
        mapper's parameters are kernel arguments, the teacher's bitfield / tables are raw pointers).  Called by
        init_pretraining(), load_checkpoint() and set_teacher(); call it after changing the teacher's mapper or occupancy
        state in place."""
        self._pt_graphs = {}
        if hasattr(self, "proxy_graph"):
            self.proxy_graph = None

    def set_teacher(self, teacher):
        """replace the teacher (a new edit): its graphs go with it"""
        self.teacher = teacher
        self.invalidate_graphs()

    def load_checkpoint(self, checkpoint, model_only=False):
        out = super().load_checkpoint(checkpoint, model_only=model_only)
        self.invalidate_graphs()  # (optimizer state tensors were replaced)
        return out

    def freeze_mlp(self, freeze=True):
        """SealNeRF/trainer.py:472-488: the NGP backbone freezes its MLPs during local pretraining; the TensoRF backbone
        (recognised by its factor lists) freezes NOTHING — its branch returns before touching a module"""
        if hasattr(self.model, "sigma_mat"):
            return
        for name in ("sigma_net", "color_net", "bg_net"):
            m = getattr(self.model, name, None)
            if m is not None:
                freeze_module(m, freeze)

    def set_lr(self, lr):
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        # an optimizer whose step has been captured once applies lr changes through a device-side factor (nerf/optim.py:
        # capture_lr): bring it up to date HERE, so that a graph replayed next — a pretraining chunk, the fine-tuning step —
        # runs at this lr and not at the factor some other phase left behind
        follow = getattr(self.optimizer, "follow_lr_schedule", None)
        if follow is not None and getattr(self.optimizer, "_lr_captured", None) is not None \
                and not torch.cuda.is_current_stream_capturing() and not follow():
            self.optimizer.capture_lr()   # groups moved by different factors: rebase, and drop every graph that baked the old lrs in
            self._pt_graphs = {}
            if getattr(self, "graph", None) is not None:
                self.graph = self.graph_opt = None

    def pretrain_loss(self, points, dirs, gt_sigma, gt_color, n_total=None):
        """SealNeRF/trainer.py:455-469: L1Loss(sigma) + L1Loss(colour) (means) of the student on one point chunk.  With a
        shard of the chunk, `n_total` is the size of the whole chunk: the shard's sums are normalised by the global count.
        `points` / `dirs` may already be padded to whole 128-row tiles (the targets say how many rows count)."""
        n = gt_sigma.shape[0]
        n_total = n_total or n
        if points.is_cuda and points.shape[0] % 128:
            # the fused network path works on whole 128-row tiles: pad; the padding rows take no part in the loss (their
            # gradient is exactly zero, so the table gradients are those of the unpadded chunk)
            pad = 128 - points.shape[0] % 128
            points, dirs = F.pad(points, (0, 0, 0, pad)), F.pad(dirs, (0, 0, 0, pad), value=1.0)
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            sigma, color = self.model(points, dirs)
            if sigma.is_cuda and self.native_optim and self.fused_losses:
                return _L1Pair.apply(sigma.reshape(-1), color.reshape(-1, 3), gt_sigma.float().contiguous(), gt_color.float().contiguous(),
                                     n_total, self._expected_grad())
            sigma, color = sigma[:n], color[:n]
            return (sigma.float() - gt_sigma).abs().sum() / n_total + (color.float() - gt_color).abs().sum() / (n_total * 3)

    def finetune_loss(self, rays_o, rays_d, gt_rgb, gt_depth=None, bg_color=1):
        """nerf/utils.py:436-537 with Seal's depth target: mean over rays of (MSE over channels + L1Loss(depth)) = MSE + L1.
        On the native path background compositing, both criteria and the backward of the MSE are one launch (nerf/trainer.py:
        _BgMse); the depth term has a value but — as in the reference, raymarching.py:274 — no gradient."""
        from nerf.trainer import render_loss
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            out = self.model.render(rays_o, rays_d, bg_color=bg_color, perturb=True, force_all_rays=False,
                                    defer_background=self.native_optim and not torch.is_tensor(bg_color),
                                    fused_loss=self._fused_loss(gt_rgb, gt_depth, self.depth_weight), **self.render_kwargs)
            # (+ the backbone trainer's own term: TensoRF's L1 penalty, tensoRF/utils.py:42-49 — the student's train_step of
            #  the reference is the backbone trainer's, SealNeRF/trainer.py:589-594)
            loss = self._regularized(render_loss(out, gt_rgb, self._expected_grad(), gt_depth, self.depth_weight))
        return loss, out

    def pretrain_step(self, points, dirs, gt_sigma, gt_color, n_total=None):
        """one optimizer step on one point shard; `n_total` = size of the un-sharded chunk (mean over all ranks)"""
        self.model.train()
        self.optimizer.zero_grad(set_to_none=False)
        world = self.dist.world if self.dist is not None else 1
        # x world because the DP layer averages the shards' gradients
        loss = self.pretrain_loss(points, dirs, gt_sigma, gt_color, n_total)
        if world != 1:
            loss = loss * world
        self._backward(loss)
        self._reduce_and_step()
        return loss.detach() / world if world != 1 else loss.detach()

    graph_pretraining = True  # GPU: every point chunk's step is replayed from its own HIP graph (static chunk tensors)
    fused_losses = True       # GPU + native optimizer: one-launch criteria (False: the reference's torch op sequences, A/B runs)

    def _pretrain_chunk(self, key, sl, n_total):
        """one optimizer step on the (static) chunk `sl` of the local points; GPU + native optimizer: captured once per chunk
        and replayed — the chunk's tensors never move, the learning rate and the frozen MLPs are part of the capture"""
        src = self.pretraining_data["local"]
        args = (src["points"][sl], src["dirs"][sl], src["sigma"][sl], src["color"][sl])
        on_gpu = args[0].is_cuda
        if on_gpu and args[0].shape[0] % 128:
            # the chunk's points and directions padded to whole 128-row tiles ONCE (they are static): pretrain_loss would pad
            # them again on every step otherwise (two fills + two copies per replay)
            cache = self.pretraining_data.setdefault("_padded", {})
            ent = cache.get(key)
            if ent is None or ent[0] != (args[0].data_ptr(), args[0].shape[0]):
                pad = 128 - args[0].shape[0] % 128
                ent = ((args[0].data_ptr(), args[0].shape[0]), F.pad(args[0], (0, 0, 0, pad)), F.pad(args[1], (0, 0, 0, pad), value=1.0))
                cache[key] = ent
            args = (ent[1], ent[2], args[2], args[3])
        if not (self.graph_pretraining and on_gpu and self.native_optim):
            return self.pretrain_step(*args, n_total=n_total)
        if not hasattr(self, "_pt_graphs"):
            self._pt_graphs = {}
        # the entry is valid for exactly the tensors it was captured on (a replaced chunk tensor or optimizer moment = re-capture)
        sig = (key, args[0].data_ptr(), args[2].data_ptr(), tuple(st["exp_avg"].data_ptr() for st in self.optimizer.state.values()
                                                                   if "exp_avg" in st), float(self.pretraining_lr))
        ent = self._pt_graphs.get(key)
        if ent is not None and ent[2] != sig:
            ent = None
        if ent is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                warm = self.pretrain_step(*args, n_total=n_total).clone()  # warm-up: a real step, this call's step
            torch.cuda.current_stream().wait_stream(side)
            world = self.dist.world if self.dist is not None else 1
            if world > 1:
                return warm  # (collective between backward and step: the steps stay eager; no entry is cached)
            g = torch.cuda.CUDAGraph()
            pool = next(iter(self._pt_graphs.values()))[0].pool() if self._pt_graphs else None
            with torch.cuda.graph(g, pool=pool, capture_error_mode=_CAPTURE_MODE):
                loss = self.pretrain_step(*args, n_total=n_total)
            self._pt_graphs[key] = (g, loss, sig)
            return warm  # (the capture itself does not execute)
        ent[0].replay()
        from gridencoder.grid import bump_weights_epoch
        bump_weights_epoch()
        return ent[1]

    def pretrain_one_epoch(self):
        """one pass over the local points (trainer.py:363-452); every rank processes its shard of each chunk"""
        from parallel import shard_slice
        if not self.model.density_bitfield_hacked:
            self.model.hack_bitfield()
        self.set_lr(self.pretraining_lr)
        self.freeze_mlp(True)
        src = self.pretraining_data["local"]
        rank, world = (self.dist.rank, self.dist.world) if self.dist is not None else (0, 1)
        losses = []
        for k, (a, b) in enumerate(zip(src["steps"][:-1], src["steps"][1:])):
            lo, hi = shard_slice(b - a, rank, world)
            losses.append(self._pretrain_chunk(k, slice(a + lo, a + hi), b - a))
        self.freeze_mlp(False)
        self.set_lr(self.base_lr)
        return losses[0] if len(losses) == 1 else torch.stack([l.reshape(()) for l in losses]).mean()

    # ------------------------------------------------------------------ global fine-tuning
    @torch.no_grad()
    def proxy_truth(self, rays_o, rays_d, out_rgb=None, out_depth=None):
        """teacher-rendered RGB + depth targets for a ray batch (force_all_rays, no perturbation) — trainer.py:506-586;
        `out_rgb` [N,3] / `out_depth` [N]: write them there (fp32, contiguous)"""
        if not self.teacher.density_bitfield_hacked:
            self.teacher.hack_bitfield()
        # The reference never puts the teacher in eval mode: its render takes run_cuda's TRAINING branch (`force_all_rays` is
        # an argument of that branch only) — one march over every ray, depth accumulated from the first sample like the
        # student's (raymarching.cu:536-552), not from the camera like the inference loop's (:844-872).  The two depth
        # conventions differ by the ray's entry distance, so targets from the inference loop would put a constant floor
        # under the L1 depth term.
        was_training = self.teacher.training
        self.teacher.train()
        native = rays_o.is_cuda and (out_rgb is not None or self.native_optim)
        try:
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
                out = self.teacher.render(rays_o, rays_d, bg_color=None, perturb=False, force_all_rays=True,
                                          defer_background=native, **self.render_kwargs)
        finally:
            self.teacher.train(was_training)
        if out.get("premultiplied", False):
            # background (nerf/renderer.py:316) + the two nan_to_num of the reference in one launch, written where the caller
            # wants the targets (the graph-replayed trainer: its static buffers)
            import s3d_hip
            n = out["weights_sum"].numel()
            rgb = out_rgb if out_rgb is not None else torch.empty(n, 3, device=rays_o.device)
            dep = out_depth if out_depth is not None else torch.empty(n, device=rays_o.device)
            bg = out["bg_color"]
            bg = (float(bg),) * 3 if not isinstance(bg, (tuple, list)) else tuple(float(v) for v in bg)
            s3d_hip.NgpHeadBackend.bg_targets(out["image"].reshape(-1, 3).float().contiguous(), out["weights_sum"].reshape(-1).float().contiguous(),
                                              out["depth"].reshape(-1).float().contiguous(), bg, rgb, dep)
            return rgb.view(out["image"].shape), dep.view(out["depth"].shape)
        rgb, dep = torch.nan_to_num(out["image"], nan=0.0), torch.nan_to_num(out["depth"], nan=0.0)
        if out_rgb is not None:
            torch._foreach_copy_([out_rgb, out_depth], [rgb.reshape(-1, 3), dep.reshape(-1)])
            return out_rgb.view(rgb.shape), out_depth.view(dep.shape)
        return rgb, dep

    def _seal_step(self, rays_o, rays_d, gt_rgb, gt_depth, bg_color=1):
        """zero grads -> fine-tuning loss -> backward -> (all-reduce) -> loss-scaled Adam, launched eagerly"""
        self.optimizer.zero_grad(set_to_none=False)
        loss, _ = self.finetune_loss(rays_o, rays_d, gt_rgb, gt_depth, bg_color)
        self._backward(loss)
        self._reduce_and_step()
        return loss.detach()


class SealTrainer(SealSteps, Trainer):
    def __init__(self, student, teacher, lr=1e-2, fp16=True, dist=None, depth_weight=1.0, **kw):
        Trainer.__init__(self, student, lr=lr, fp16=fp16, dist=dist, **kw)
        self._init_seal(teacher, lr, depth_weight)

    def train_step(self, rays_o, rays_d, gt_rgb=None, gt_depth=None, bg_color=1):
        if gt_rgb is None:
            gt_rgb, gt_depth = self.proxy_truth(rays_o, rays_d)
        self.model.train()
        self._maybe_update_extra_state()  # (with data parallelism: occupancy state re-synchronised over the ranks)
        self.global_step += 1
        return self._seal_step(rays_o, rays_d, gt_rgb, gt_depth, bg_color)


def _tensorf_seal_trainer():
    from tensoRF.utils import TensoRFSteps

    class SealTensoRFTrainer(SealSteps, TensoRFSteps, Trainer):
        """the student trainer of main_SealTensoRF.py (`get_trainer(BackBoneTypes.TensoRF, CharacterTypes.Student)`,
        SealNeRF/trainer.py:56-104): Seal's steps on top of the TensoRF trainer — two learning rates, the L1 penalty on the
        density factors inside every fine-tuning step, nothing frozen during local pretraining (freeze_mlp)"""

        def __init__(self, student, teacher, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True, dist=None, depth_weight=1.0, **kw):
            self._init_tensorf(lr0, lr1, l1_reg_weight, kw.pop("upsample_model_steps", ()), kw.pop("upsample_resolutions", ()))
            Trainer.__init__(self, student, lr=lr0, fp16=fp16, dist=dist, **kw)
            self._init_seal(teacher, lr0, depth_weight)
            self._attach_source_checks()

        def train_step(self, rays_o, rays_d, gt_rgb=None, gt_depth=None, bg_color=1):
            loss = SealTrainer.train_step(self, rays_o, rays_d, gt_rgb, gt_depth, bg_color)
            self._maybe_upsample()
            return loss
    return SealTensoRFTrainer


def _tensorf_seal_graphed_trainer():
    from tensoRF.utils import TensoRFSteps

    class SealTensoRFGraphedTrainer(TensoRFSteps, GraphedSealTrainer):
        """the same student trainer with the fine-tuning step replayed from a HIP graph (GraphedSealTrainer); the teacher's proxy
        render stays eager on this backbone (its kernels take no device-side row count: `honours_row_limit_under_autocast`), and a
        captured step is dropped when the resolution schedule changes the factors"""

        def __init__(self, student, teacher, num_rays, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, fp16=True, dist=None, depth_weight=1.0, **kw):
            self._init_tensorf(lr0, lr1, l1_reg_weight, kw.pop("upsample_model_steps", ()), kw.pop("upsample_resolutions", ()))
            kw.setdefault("budget_factor", 1.1)  # (tensoRF/utils.py: GraphedTrainer)
            GraphedSealTrainer.__init__(self, student, teacher, num_rays, lr=lr0, fp16=fp16, dist=dist, depth_weight=depth_weight, **kw)
            self._attach_source_checks()

        def train_step(self, rays_o, rays_d, gt_rgb=None, gt_depth=None, bg_color=1):
            loss = GraphedSealTrainer.train_step(self, rays_o, rays_d, gt_rgb, gt_depth, bg_color)
            self._maybe_upsample()
            return loss
    return SealTensoRFGraphedTrainer


def get_trainer(backbone="ngp", graphed=False):
    """SealNeRF/trainer.py:56-104 `get_trainer(backbone, CharacterTypes.Student)`: the student trainer class of a backbone
    ("ngp": nerf/network.py or network_ff.py, "tensorf": tensoRF/network.py)"""
    if backbone == "tensorf":
        return _tensorf_seal_graphed_trainer() if graphed else _tensorf_seal_trainer()
    if backbone != "ngp":
        raise ValueError(f"unknown backbone {backbone!r}")
    return GraphedSealTrainer if graphed else SealTrainer


class GraphedSealTrainer(SealSteps, GraphedTrainer):
    """Seal fine-tuning with the student's step replayed from a HIP graph (nerf.trainer.GraphedTrainer): the teacher's proxy
    render produces the targets eagerly (its sample count is data dependent and it takes no gradient), they are copied into
    static buffers, and zero-grad -> march -> two encoders -> MFMA MLPs -> composite -> MSE + L1(depth) -> backward ->
    loss-scaled Adam is one replay.  Pretraining uses SealTrainer's per-chunk graphs."""

    def __init__(self, student, teacher, num_rays, lr=1e-2, fp16=True, dist=None, depth_weight=1.0, **kw):
        GraphedTrainer.__init__(self, student, num_rays, lr=lr, fp16=fp16, dist=dist, **kw)
        self._init_seal(teacher, lr, depth_weight)
        self.s_depth = torch.zeros(num_rays, device=self.s_ro.device)
        self.proxy_graph = None

    graph_proxy = True  # the teacher's proxy render replayed from its own HIP graph (static rays in, static targets out)

    def _proxy_replay(self):
        """teacher proxy render of the rays in s_ro / s_rd into s_gt / s_depth.  The render takes run_cuda's un-budgeted
        training branch (force_all_rays): N * max_steps sample rows of static extent, the real count on the device (every
        kernel of the two-encoder network takes it as n_valid) — no host read-back, so it can be captured."""
        def body():
            self.proxy_truth(self.s_ro, self.s_rd, self.s_gt, self.s_depth)
        if self.proxy_graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                body()  # warm-up = this call's render
            torch.cuda.current_stream().wait_stream(side)
            self.proxy_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.proxy_graph, capture_error_mode=_CAPTURE_MODE):
                body()
            return
        self.proxy_graph.replay()

    def _static_loss(self):
        loss, _ = self.finetune_loss(self.s_ro, self.s_rd, self.s_gt, self.s_depth, bg_color=1)
        return loss

    def _stage_inputs(self, rays_o, rays_d, gt_rgb):
        gt_rgb, gt_depth = gt_rgb
        if gt_rgb is self.s_gt:  # targets (and rays) were staged by the proxy graph's caller
            return
        torch._foreach_copy_([self.s_ro, self.s_rd, self.s_gt, self.s_depth],
                             [rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), gt_rgb.reshape(-1, 3), gt_depth.reshape(-1)])

    def _eager_step(self, rays_o, rays_d, gt, bg_color=1):
        return self._seal_step(rays_o, rays_d, gt[0], gt[1], bg_color)

    def train_step(self, rays_o, rays_d, gt_rgb=None, gt_depth=None, bg_color=1):
        if gt_rgb is None:
            ok = (self.graph_proxy and self.fp16 and rays_o.is_cuda and rays_o.numel() == self.s_ro.numel()
                  and self.teacher.honours_row_limit_under_autocast(rays_o.numel() // 3 * self.render_kwargs["max_steps"]))
            if ok:
                torch._foreach_copy_([self.s_ro, self.s_rd], [rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)])
                self._proxy_replay()
                gt_rgb, gt_depth = self.s_gt, self.s_depth
            else:
                gt_rgb, gt_depth = self.proxy_truth(rays_o, rays_d)
        return GraphedTrainer.train_step(self, rays_o, rays_d, (gt_rgb, gt_depth), bg_color)
