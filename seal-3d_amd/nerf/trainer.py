"""Minimal NGP training step (the hot-path part of nerf/utils.py `Trainer`): rays -> render -> MSE -> backward ->
Adam, with fp16 autocast + GradScaler (`-O`), `update_extra_state` every 16 steps (nerf/utils.py:845-848),
Adam(betas=(0.9, 0.99), eps=1e-15) and lr 1e-2 (main_SealNeRF.py:283-288), PSNR as in nerf/utils.py:226-233.
Optional data parallelism over rays: gradients are all-reduced through one flat bucket (parallel/dist.py)."""
import math

import torch
import torch.nn.functional as F

from gridencoder.grid import bump_weights_epoch


def psnr(pred, target):
    return -10 * math.log10(float(torch.mean((pred.float() - target.float()) ** 2)) + 1e-20)


class _BgMse(torch.autograd.Function):
    """loss = mean((image + (1 - weights_sum) * bg - gt)^2): background compositing (nerf/renderer.py:316) and the MSE
    criterion (nerf/utils.py:484) in one kernel per direction (csrc/ngp_head.hip) instead of ~14 tiny launches.  With
    `depth` / `gt_depth`: + depth_weight * L1Loss(nan_to_num(depth), gt_depth), Seal-3D's depth term (nerf/utils.py:486-489)
    — a value only: the reference's composite backward drops the depth gradient (raymarching.py:274)."""

    @staticmethod
    def forward(ctx, image, weights_sum, gt, bg, expected_grad=None, depth=None, gt_depth=None, depth_weight=1.0):
        """`expected_grad`: the (device scalar) upstream gradient the loss WILL receive — the loss scale under GradScaler; the
        forward launch then writes the backward's result as well and backward() launches nothing when it is handed that tensor"""
        import s3d_hip
        image, weights_sum, gt = image.float().contiguous(), weights_sum.float().contiguous(), gt.float().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=image.device)
        ctx.pre = None
        extra = {}
        if depth is not None:
            extra = dict(depth=depth.detach().float().contiguous().reshape(-1), gt_depth=gt_depth.float().contiguous().reshape(-1),
                         depth_weight=float(depth_weight))
        if expected_grad is not None and expected_grad.dtype == torch.float32 and expected_grad.numel() == 1:
            g_image, g_ws = torch.empty_like(image), torch.empty_like(weights_sum)
            s3d_hip.NgpHeadBackend.bg_mse_forward(image, weights_sum, gt, bg, loss, expected_grad, g_image, g_ws, **extra)
            ctx.pre = (g_image, g_ws, expected_grad.data_ptr(), expected_grad._version)
        else:
            s3d_hip.NgpHeadBackend.bg_mse_forward(image, weights_sum, gt, bg, loss, **extra)
        ctx.save_for_backward(image, weights_sum, gt)
        ctx.bg = bg
        return loss

    @staticmethod
    def backward(ctx, g):
        import s3d_hip
        if ctx.pre is not None and g.dtype == torch.float32 and g.data_ptr() == ctx.pre[2] and g._version == ctx.pre[3]:
            return (ctx.pre[0], ctx.pre[1]) + (None,) * 6  # (the announced upstream gradient: already computed)
        image, weights_sum, gt = ctx.saved_tensors
        g_image, g_ws = torch.empty_like(image), torch.empty_like(weights_sum)
        s3d_hip.NgpHeadBackend.bg_mse_backward(image, weights_sum, gt, ctx.bg, g.float().contiguous(), g_image, g_ws)
        return (g_image, g_ws) + (None,) * 6


def render_loss(out, gt_rgb, expected_grad=None, gt_depth=None, depth_weight=1.0):
    """MSE between the rendered batch and the targets (+ Seal-3D's L1 depth term when `gt_depth` is given); uses the fused
    kernel when the renderer deferred the background (`expected_grad`: see _BgMse.forward)"""
    if "loss" in out:  # the renderer's compositing launch formed the criterion itself (Trainer._fused_loss -> render(fused_loss=...))
        return out["loss"]
    if out.get("premultiplied", False):
        bg = out["bg_color"]
        bg = (float(bg),) * 3 if not isinstance(bg, (tuple, list)) else tuple(float(v) for v in bg)
        if gt_depth is not None:
            return _BgMse.apply(out["image"].reshape(-1, 3), out["weights_sum"].reshape(-1), gt_rgb.reshape(-1, 3), bg, expected_grad,
                                out["depth"], gt_depth, depth_weight)
        return _BgMse.apply(out["image"].reshape(-1, 3), out["weights_sum"].reshape(-1), gt_rgb.reshape(-1, 3), bg, expected_grad)
    loss = F.mse_loss(out["image"], gt_rgb)
    if gt_depth is not None:
        loss = loss + depth_weight * F.l1_loss(torch.nan_to_num(out["depth"], nan=0.0).view(gt_depth.shape), gt_depth)
    return loss


class Trainer:
    def __init__(self, model, lr=1e-2, fp16=True, update_extra_interval=16, dist=None, max_steps=1024, dt_gamma=0,
                 T_thresh=1e-4, capturable=False, native_optim=None, optimizer=None, scaler=None, lr_scheduler=None):
        self.model = model
        # `lr_scheduler(optimizer)` -> a torch scheduler, stepped after every optimizer step (nerf/utils.py:411-414 with
        # `scheduler_update_every_step=True`, main_SealNeRF.py:283-300: LambdaLR 0.1 ** min(iter / iters, 1)); a replayed step
        # follows it through the optimizer's device-side lr factor (nerf/optim.py: follow_lr_schedule)
        self._lr_scheduler_factory = lr_scheduler
        self.lr_scheduler = None
        self.lr = lr
        self.fp16 = fp16
        self.update_extra_interval = update_extra_interval
        self.dist = dist
        self.render_kwargs = dict(max_steps=max_steps, dt_gamma=dt_gamma, T_thresh=T_thresh)
        on_gpu = next(model.parameters()).is_cuda
        # GPU default: Adam + loss scaling straight from the (fp16) gradients of the HIP kernels (nerf/optim.py);
        # native_optim=False keeps the reference's torch.optim.Adam + GradScaler (the only choice on CPU)
        self.native_optim = on_gpu if native_optim is None else (native_optim and on_gpu)
        self._capturable = capturable and on_gpu
        if optimizer is not None:  # share an existing optimizer / scaler (e.g. an eager twin of a graphed trainer)
            self.optimizer, self.scaler = optimizer, scaler
            if lr_scheduler is not None:
                self.lr_scheduler = lr_scheduler(self.optimizer)
        else:
            self.scaler = None
            self.rebuild_optimizer()
        self.global_step = 0
        self.epoch = 0
        self.stats = {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}
        if dist is not None:
            dist.register(model)  # broadcasts rank 0's parameters through `.data` (no version bump) ...
            resync = getattr(self.optimizer, "resync_half", None)
            if resync is not None:
                resync()  # ... so the optimizer's fp16 copies are re-made from the adopted values
            bump_weights_epoch()

    def _param_groups(self):
        """the optimizer's parameter groups (a backbone's trainer may use several learning rates: tensoRF/utils.py)"""
        return self.model.get_params(self.lr)

    def rebuild_optimizer(self):
        """(re-)create optimizer (and, the first time, the loss scaler) over the model's CURRENT parameters — needed after
        the parameter set changes (TensoRF upsample_model, tensoRF/utils.py:137-140 / :347-352)"""
        model, lr, fp16 = self.model, self.lr, self.fp16
        on_gpu = next(model.parameters()).is_cuda
        if self.native_optim:
            from .optim import NativeAdam, NativeGradScaler
            self.optimizer = NativeAdam(self._param_groups(), lr=lr, betas=(0.9, 0.99), eps=1e-15)
            if self.scaler is None:
                self.scaler = NativeGradScaler(next(model.parameters()).device, enabled=fp16)
            if self.dist is None:
                self.scaler.attach(self.optimizer)
        else:
            self.optimizer = torch.optim.Adam(self._param_groups(), betas=(0.9, 0.99), eps=1e-15, fused=on_gpu,
                                              capturable=self._capturable)
            if self.scaler is None:
                self.scaler = torch.amp.GradScaler("cuda", enabled=fp16)
        if self._lr_scheduler_factory is not None:
            self.lr_scheduler = self._lr_scheduler_factory(self.optimizer)

    def _sched_step(self):
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()

    def save_checkpoint(self, workspace, name="ngp", full=False, best=False, remove_old=True, max_keep_ckpt=2):
        """reference on-disk format (nerf/utils.py:1015-1076); see nerf/checkpoint.py"""
        from .checkpoint import save_checkpoint
        return save_checkpoint(self, workspace, name, full=full, best=best, remove_old=remove_old, max_keep_ckpt=max_keep_ckpt)

    def load_checkpoint(self, checkpoint, model_only=False):
        """reference on-disk format (nerf/utils.py:1078-1137); a file written by the reference's Trainer loads as is"""
        from .checkpoint import load_checkpoint
        return load_checkpoint(self, checkpoint, model_only=model_only)

    def _maybe_update_extra_state(self):
        model = self.model
        if model.cuda_ray and self.global_step % self.update_extra_interval == 0:
            with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
                model.update_extra_state()
            if self.dist is not None:
                # replicas draw different random cells: adopt rank 0's grid / bitfield / mean_count so that every rank
                # marches the same occupancy and takes the same (re-)capture decisions below
                self.dist.sync_extra_state(model)
            return True
        return False

    def _eager_step(self, rays_o, rays_d, gt_rgb, bg_color=1):
        model = self.model
        # without data parallelism the gradients are simply replaced each step (no 49 MB zero fill + 147 MB accumulate);
        # with it they live in the flat all-reduce bucket and are cleared in place
        self.optimizer.zero_grad(set_to_none=self.dist is None)
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            out = model.render(rays_o, rays_d, bg_color=bg_color, perturb=True, force_all_rays=False,
                               defer_background=self.native_optim, fused_loss=self._fused_loss(gt_rgb), **self.render_kwargs)
            loss = self._regularized(render_loss(out, gt_rgb, self._expected_grad()))
        self._backward(loss)
        self._reduce_and_step()
        return loss.detach()

    fused_losses = True  # GPU + native optimizer: one-launch criteria (False: the unfused sequences, A/B runs and parity tests)

    def _fused_loss(self, gt_rgb, gt_depth=None, depth_weight=1.0):
        """`fused_loss=` of NeRFRenderer.run_cuda: targets and the loss's announced upstream gradient for
        raymarching.composite_rays_train_loss — compositing, criterion and compositing backward as one launch.
        None where the step has no announced gradient (torch's GradScaler) or the fusion is switched off."""
        g = self._expected_grad()
        if g is None or not self.fused_losses or not self.native_optim or not gt_rgb.is_cuda:
            return None
        return dict(gt=gt_rgb, expected_grad=g, gt_depth=gt_depth, depth_weight=depth_weight)

    def _regularizer(self):
        """extra loss term of a backbone's trainer (TensoRF: `density_loss() * l1_reg_weight`, tensoRF/utils.py:42-49); None: none"""
        return None

    def _regularized(self, loss):
        reg = self._regularizer()
        return loss if reg is None else loss + reg.to(loss.dtype)

    def _reduce_and_step(self):
        """gradient all-reduce (data parallelism) + optimizer step.  Native optimizer: the reduction is issued in pieces and
        pipelined with the per-parameter updates (NativeGradScaler.step); torch optimizer: one reduction, then the step."""
        if self.dist is not None and self.native_optim:
            self.scaler.step(self.optimizer, dist=self.dist)
        else:
            if self.dist is not None:
                self.dist.allreduce_grads(self.scaler)
            self.scaler.step(self.optimizer)
        self.scaler.update()

    def _expected_grad(self):
        """the upstream gradient `_backward` will hand to the loss (NativeGradScaler: its scale tensor), if known in advance"""
        sc = self.scaler
        return sc._scale.reshape(()) if (hasattr(sc, "backward") and getattr(sc, "enabled", False)) else None

    # the hash tables' Adam inside their backward's accumulate kernel (nerf/optim.py: arm_fused_tables); S3D_FUSE_TABLE_ADAM=0: A/B
    fuse_table_updates = __import__("os").environ.get("S3D_FUSE_TABLE_ADAM", "1") != "0"

    def _arm_fused_tables(self):
        """single replica, native optimizer + scaler, every gradient of the step a hand-over buffer whose producer raises the
        scaler's flag itself: the step's skip decision is complete when the tables' backward — the last node of the graph —
        starts, so their update can be applied there.  (Seal's nn.Linear `.grad`s without a pack, TensoRF's factors, data
        parallelism: the separate update as before.)"""
        opt, sc = self.optimizer, self.scaler
        if not (self.fuse_table_updates and self.native_optim and self.dist is None and hasattr(opt, "arm_fused_tables")
                and hasattr(sc, "_checked_at_source")):
            return 0
        params = [p for g in opt.param_groups for p in g["params"]]
        if not all(getattr(p, "_s3d_grad", None) is not None for p in params) or not sc._checked_at_source(opt):
            return 0
        return opt.arm_fused_tables(grad_scale=sc._scale if sc.enabled else None)

    def _backward(self, loss):
        """the step's ONE backward pass (every caller follows it with `_reduce_and_step`)"""
        self._arm_fused_tables()
        if hasattr(self.scaler, "backward"):  # NativeGradScaler: the scale is passed as the root gradient
            self.scaler.backward(loss)
        else:
            self.scaler.scale(loss).backward()

    def train_step(self, rays_o, rays_d, gt_rgb, bg_color=1):
        """rays_o/d [N,3], gt_rgb [N,3].  Returns the (detached) loss tensor; no host sync."""
        self.model.train()
        self._maybe_update_extra_state()
        self.global_step += 1
        loss = self._eager_step(rays_o, rays_d, gt_rgb, bg_color)
        self._sched_step()
        return loss

    @torch.no_grad()
    def render_image(self, rays_o, rays_d, bg_color=1):
        self.model.eval()
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            return self.model.render(rays_o, rays_d, bg_color=bg_color, perturb=False, **self.render_kwargs)


# Stream capture restricts HIP calls of the CAPTURING thread only: with a process group alive, torch's RCCL watchdog thread
# polls its work events (hipEventQuery) at any time, and under the default "global" mode such a poll during one of the
# captures below aborts the process ("operation not permitted when stream is capturing").
_CAPTURE_MODE = "thread_local"
_RING_PUSH = __import__("os").environ.get("S3D_RING_PUSH", "1") != "0"  # A/B switch: 0 = host-side ring bookkeeping


class GraphedTrainer(Trainer):
    """The same training step replayed from a HIP graph (torch.cuda.CUDAGraph): the step issues ~115 kernels whose
    launch cost (~0.3 ms of a 2.5 ms step, rocprof: GPU 79 % busy) is paid once at capture.

    What makes the step capturable: no host sync inside it (the sample budget M is a *static* allocation instead of the
    16-step running mean — rays that do not fit are dropped exactly as in the reference, raymarching.cu:416 — fused
    Adam + GradScaler keep found_inf on the device), static input buffers, libseal3d_hip launches on the capture
    stream.  `update_extra_state` (data-dependent shapes, `.item()`) stays eager every 16 steps; the graph is
    re-captured only when the budget has to grow.

    The budget is generous (`budget_factor` x the running mean at capture): every per-sample kernel of the step takes the
    march's device-side sample count (`n_valid`, seal3d_hip.h) and skips the unused tail of the buffers, so a larger M
    costs two fills, not compute — fewer dropped rays than the reference's M = running mean, and re-captures only when
    the mean outgrows the headroom."""

    def __init__(self, model, num_rays, budget_factor=1.3, graph_extra_state=True, capture_collectives=True, **kw):
        super().__init__(model, capturable=True, **kw)
        self.capture_collectives = capture_collectives  # False: two graphs with the all-reduce issued eagerly between them
        self.collectives_in_graph = False
        dev = next(model.parameters()).device
        self.s_ro = torch.zeros(num_rays, 3, device=dev)
        self.s_rd = torch.zeros(num_rays, 3, device=dev)
        self.s_gt = torch.zeros(num_rays, 3, device=dev)
        self.budget_factor = budget_factor
        self.graph_extra_state = graph_extra_state
        self.ues_graph, self.ues_mean, self.ues_warm = None, None, False
        self.graph = None
        self.graph_opt = None
        self.n_captures = 0
        self.budget = 0
        self.s_loss = None
        self.s_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        # the graph itself files each step's loss and sample counter in 16-slot rings (seal3d_hip.h: s3d_step_ring_push)
        # (the loss history is longer than the counter ring: the tensor train_step() hands out for step k is a view of slot
        #  k % 1024 and stays valid for the next 1,023 steps — a caller that keeps loss tensors longer clones them)
        self.loss_ring = torch.zeros(1024, dtype=torch.float32, device=dev)
        self._pushes = 0  # ring pushes executed so far = the device's running step number (s_cursor[1])
        self.s_cursor = torch.zeros(2, dtype=torch.int32, device=dev)  # {ring slot, running step number}
        self.noise_key = (torch.initial_seed() * 0x9E3779B1 + (self.dist.rank if self.dist is not None else 0) * 0x85EBCA6B) & 0xFFFFFFFF
        self._counter_ring = None
        self._ring_args = None

    def _static_loss(self):
        """the step's loss on the static input buffers (subclasses: other criteria, e.g. Seal's depth term)"""
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            out = self.model.render(self.s_ro, self.s_rd, bg_color=1, perturb=True, force_all_rays=False,
                                    defer_background=self.native_optim, fused_loss=self._fused_loss(self.s_gt), **self.render_kwargs)
            return self._regularized(render_loss(out, self.s_gt, self._expected_grad()))

    def _body_fb(self):
        """zero grads -> render -> loss -> scaled backward"""
        # without data parallelism the gradients are simply replaced each step (no 49 MB zero fill + 147 MB accumulate);
        # with it they live in the flat all-reduce bucket and are cleared in place
        self.optimizer.zero_grad(set_to_none=self.dist is None)
        loss = self._static_loss()
        self._backward(loss)
        loss = loss.detach()
        self._ring_args = None
        if self._counter_ring is not None:
            args = (loss.float().reshape(()), self.s_counter, self.loss_ring, self._counter_ring, self.s_cursor)
            if self.native_optim:
                self._ring_args = args  # rides in the scaler update's launch (_body_opt)
            else:
                import s3d_hip
                s3d_hip.OptimBackend.step_ring_push(*args)
        return loss

    def _body_opt(self):
        self.scaler.step(self.optimizer)
        if self._ring_args is not None:
            self.scaler.update(ring_push=self._ring_args)
            self._ring_args = None
        else:
            self.scaler.update()

    def _capture(self):
        """One graph for the whole step — with data parallelism too when the process group's collectives can be captured
        (RCCL; probed once, parallel/dist.py: capture_supported).  Otherwise (gloo, `capture_collectives=False`) the step
        is captured as TWO graphs (forward+backward | check+Adam+scaler update) and the gradient all-reduce is issued eagerly
        between the two replays."""
        model = self.model
        model.train()
        self.n_captures += 1
        if hasattr(self.optimizer, "capture_lr"):
            self.optimizer.capture_lr()  # (this lr becomes a launch argument; later values reach the replay through lr_scale)
        self._graph_lr_epoch = getattr(self.optimizer, "lr_epoch", 0)
        self.budget = int(max(model.mean_count, 1) * self.budget_factor)
        saved = (model.mean_count, model.local_step)
        model.mean_count = self.budget
        # the marcher always counts in a private buffer; the last kernel of the step files it in the ring slot of a device
        # cursor (which follows `local_step % 16`), and leaves the private counter cleared for the next replay
        ring = model.step_counter
        self._counter_ring = ring if ring.is_contiguous() and ring.dtype == torch.int32 and _RING_PUSH else None
        self.s_counter.zero_()
        self.s_cursor[0].fill_(saved[1] % 16)
        model.step_counter = self.s_counter.view(1, 2).expand(16, 2)
        model._counter_prezeroed = self._counter_ring is not None
        if self._counter_ring is not None:  # per-ray jitter drawn from the device-side step number instead of torch.rand
            model._noise_step, model._noise_key = self.s_cursor[1:], self.noise_key
        model.local_step = 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # ONE real step on the current batch, launched eagerly on a side stream: it is this call's training step (the
            # capture below records without executing, and train_step does not replay after a capture), and it warms the
            # allocator pools the capture will draw from (the eager steps before the first capture did the rest); every
            # rank runs the same number of all-reduces
            model.local_step = 0
            self.s_warm_loss = self._body_fb().clone()
            if self.dist is not None:
                self.dist.allreduce_grads(self.scaler)
            self._body_opt()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        model.local_step = 0
        if self.dist is None:
            with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
                self.s_loss = self._body_fb()
                self._body_opt()
            self.graph_opt = None
        elif self.capture_collectives and self.dist.capture_supported():
            # data parallelism, ONE graph: forward + backward, the gradient all-reduce (RCCL records into the capture: its
            # kernels run on the process group's stream, forked from and joined back into the step's stream by event edges),
            # check + Adam + scale update.  One collective per gradient buffer — every fork / join edge costs ~9 us here.
            saved_chunk = self.dist.chunk_bytes
            self.dist.chunk_bytes = 1 << 40
            try:
                with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
                    self.s_loss = self._body_fb()
                    self.dist.allreduce_grads(self.scaler)
                    self._body_opt()
            finally:
                self.dist.chunk_bytes = saved_chunk
            self.graph_opt = None
            self.collectives_in_graph = True
        else:
            with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
                self.s_loss = self._body_fb()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), capture_error_mode=_CAPTURE_MODE):
                self._body_opt()
            self.collectives_in_graph = False
        model.step_counter = ring
        model._counter_prezeroed = False
        model._noise_step = None
        model.mean_count, model.local_step = saved

    def load_checkpoint(self, checkpoint, model_only=False):
        out = super().load_checkpoint(checkpoint, model_only=model_only)
        self._counter_ring = None
        self.graph = self.graph_opt = None  # optimizer state tensors were replaced, mean_count may have moved: re-capture
        return out

    def _maybe_update_extra_state(self):
        """steady state: the partial occupancy update replayed from its own HIP graph (~30 launches, two host syncs
        less), then one host read for mean density / mean sample count.  First sweeps, models that extend
        `update_extra_state`, and `graph_extra_state=False` keep the eager reference sequence."""
        from .renderer import NeRFRenderer
        model = self.model
        if not (model.cuda_ray and self.global_step % self.update_extra_interval == 0):
            return False
        # (a model may extend update_extra_state by a pure epilogue — the Seal renderers re-mark the edit region in the
        #  bitfield, SealNeRF/renderer.py:50-66 — and say so: `after_extra_state`)
        epilogue = getattr(model, "after_extra_state", None) if getattr(type(model), "extra_state_epilogue_only", False) else None
        plain = type(model).update_extra_state is NeRFRenderer.update_extra_state or epilogue is not None
        if not (self.graph_extra_state and plain and model.iter_density >= 16) or getattr(model, "dist_shard", None) is not None:
            return super()._maybe_update_extra_state()  # (sharded over the ranks: density queries split + all-gather)
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16):
            if self.ues_graph is None and not self.ues_warm:
                mean = model.partial_grid_update_device()  # first time: eager (lazy initialisations, allocator warm-up)
                self.ues_warm = True
            else:
                if self.ues_graph is None:
                    self.ues_graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.ues_graph, capture_error_mode=_CAPTURE_MODE):
                        self.ues_mean = model.partial_grid_update_device()
                self.ues_graph.replay()
                mean = self.ues_mean
            model.finish_extra_state(mean)
            if epilogue is not None:
                epilogue()
        if self.dist is not None:
            self.dist.sync_extra_state(model)
        return True

    def _stage_inputs(self, rays_o, rays_d, gt_rgb):
        torch._foreach_copy_([self.s_ro, self.s_rd, self.s_gt],
                             [rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), gt_rgb.reshape(-1, 3)])  # one launch

    def _replay(self):
        # an eager backward without a step since the last replay left gradients in the hand-over buffer that the captured
        # step (recorded with a clean buffer: no fill) would accumulate onto: clear them first (host-side flags, no sync)
        dirty = getattr(self.optimizer, "clear_unconsumed", None)
        if dirty is not None:
            dirty()
        self.graph.replay()
        if self.graph_opt is not None:
            self.dist.allreduce_grads(self.scaler)
            self.graph_opt.replay()
        # the replay files its loss at slot s_cursor[1] % loss_slots and advances the device cursor: the host count moves HERE,
        # next to the launch, so that any caller of _replay() keeps the two in step
        slot = self._pushes % self.loss_ring.numel()
        if self._counter_ring is not None:
            self._pushes += 1
        return slot

    def train_step(self, rays_o, rays_d, gt_rgb, bg_color=1):
        model = self.model
        model.train()
        if self._maybe_update_extra_state():
            self.s_cursor[0].zero_()  # (update_extra_state restarts the counter ring: local_step = 0)
            if self.graph is not None and (model.mean_count * 1.1 > self.budget or model.mean_count * 2 < self.budget):
                self.graph = None  # the running mean left the static budget's useful range: re-capture
        self.global_step += 1
        if self.graph is None and model.mean_count <= 0:
            # no sample statistics yet (first 16 steps): eager step with the wrapper's host sync
            loss = self._eager_step(rays_o, rays_d, gt_rgb, bg_color)
            self._sched_step()
            return loss
        if bg_color != 1:
            raise ValueError("GraphedTrainer: the captured step composites on the white background (bg_color=1) of the "
                             "BASELINE configs; use Trainer for per-batch background colours")
        self._stage_inputs(rays_o, rays_d, gt_rgb)
        if self.graph is not None and hasattr(self.optimizer, "follow_lr_schedule") and not self.optimizer.follow_lr_schedule():
            self.graph = None  # (parameter groups moved by different factors: the captured lr arguments no longer fit)
        if self.graph is not None and getattr(self.optimizer, "lr_epoch", 0) != getattr(self, "_graph_lr_epoch", 0):
            self.graph = None  # (somebody rebased the captured lrs since — another graph's capture, an eager step: re-capture)
        if self.graph is None:
            self._capture()  # runs this step eagerly (one optimizer update), then records the graph
            loss = self.s_warm_loss
            filed = self._counter_ring is not None
            self._pushes += 1 if filed else 0  # (the warm-up step filed its loss too)
        else:
            slot = self._replay()
            filed = self._counter_ring is not None
            # the static loss buffer is overwritten by the next replay; the graph files each step's loss in a 1,024-slot
            # history (no extra launch per step): the caller gets the VIEW of this step's slot, untouched for 1,023 more steps
            # — a caller that keeps loss tensors longer than that (statistics over an epoch) clones or reads them
            loss = self.loss_ring[slot] if filed else self.s_loss.clone()
        bump_weights_epoch()  # replays update the parameters without touching Tensor._version
        if not filed:
            model.step_counter[model.local_step % 16].copy_(self.s_counter)
        model.local_step += 1
        self._sched_step()
        return loss
