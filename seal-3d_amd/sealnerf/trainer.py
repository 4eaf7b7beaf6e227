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
        self._pt_graphs, self._pt_padded = {}, {}
        self._cached_lr, self._pretraining_open, self.is_pretraining = None, False, False
        self.cache_gt = False

    # ------------------------------------------------------------------ local pretraining
    def _teacher_query(self, points, dirs):
        dev = points.device
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.fp16 and dev.type == "cuda"):
            chunks = [self.teacher(points[i:i + (1 << 20)], dirs[i:i + (1 << 20)]) for i in range(0, points.shape[0], 1 << 20)]
        if not chunks:
            return torch.zeros(0, device=dev), torch.zeros(0, 3, device=dev)
        return torch.cat([c[0].float() for c in chunks]), torch.cat([c[1].float() for c in chunks])

    @torch.no_grad()
    def init_pretraining(self, epochs=0, batch_size=6144000, lr=0.05, local_point_step=0.005, local_angle_step=45,
                         surrounding_point_step=-1, surrounding_angle_step=45, surrounding_bounds_extend=0.2,
                         global_point_step=-1, global_angle_step=45, seed=0):
        """SealNeRF/trainer.py:88-263: the three point sets of the distillation pretraining, each with teacher targets —
          local        lattice inside the force-fill bounds, kept where the proxy maps (everywhere with `mapSource`);
                       targets = teacher at the MAPPED points for the mapped constant direction (1,0,0), colours through the edit
          surrounding  lattice inside the bounds grown by `surrounding_bounds_extend` (clamped to the training box; as in the
                       reference the mapper's `force_fill_bound` is grown IN PLACE, :166-180), kept where the proxy does NOT map
          global       lattice over the training box, kept where the proxy does not map
        A part is skipped when its step is <= 0 (main_SealNeRF.py:93-108 defaults: local 0.001, surrounding 0.01 / extend 0.1,
        global off).  The training directions are drawn per kept point from the Euler grid; `seed=None` draws them from torch's
        global generator exactly where the reference does (one randint per part), an int seeds a private generator (every
        data-parallel rank holds the same set).  Returns the number of local points."""
        mapper = self.teacher.seal_mapper
        dev = next(self.model.parameters()).device
        self.pretraining_epochs, self.pretraining_batch_size, self.pretraining_lr = epochs, batch_size, lr
        self.pretraining_data = {}
        gen = None if seed is None else torch.Generator(device="cpu").manual_seed(seed)
        axis = torch.tensor([1.0, 0, 0], device=dev)

        def lattice(bounds, point_step, angle_step):
            pts, dirs = sample_points(bounds, point_step, angle_step)
            pts, dirs = pts.to(dev, torch.float32), dirs.to(dev, torch.float32)
            return (pts, dirs) + tuple(mapper.map_to_origin(pts, torch.zeros_like(pts) + axis))

        def draw(dirs, n):
            return dirs[torch.randint(dirs.shape[0], (n,), generator=gen).to(dev)]

        def file(part, pts, dirs, sigma, color):
            steps = list(range(0, pts.shape[0], batch_size))
            if not steps or steps[-1] != pts.shape[0]:
                steps.append(pts.shape[0])
            self.pretraining_data[part] = {"points": pts.contiguous(), "dirs": dirs.contiguous(), "sigma": sigma.detach().contiguous(),
                                           "color": color.detach().contiguous(), "steps": steps}

        n_local = 0
        if local_point_step > 0:
            pts, dirs, mapped_p, mapped_d, mask = lattice(mapper.map_data["force_fill_bound"], local_point_step, local_angle_step)
            if "map_source" in mapper.map_data:
                mask = torch.ones_like(mask)
            pts = pts[mask]
            dirs = draw(dirs, pts.shape[0])
            mapped_p, mapped_d = mapped_p[mask], mapped_d[mask]
            gt_sigma, gt_color = self._teacher_query(mapped_p, mapped_d)
            file("local", pts, dirs, gt_sigma, mapper.map_color(mapped_p, mapped_d, gt_color))
            n_local = pts.shape[0]
            self.is_pretraining = True
        if surrounding_point_step > 0:
            sb = mapper.map_data["force_fill_bound"]
            aabb = self.model.aabb_train.to(sb.device)
            lo, hi = (sb[0], sb[1]) if sb.ndim == 2 else (sb[:, 0], sb[:, 1])  # (views: the growth lands in map_data)
            lo -= surrounding_bounds_extend
            lo.copy_(torch.max(lo, aabb[:3]))
            hi += surrounding_bounds_extend
            hi.copy_(torch.min(hi, aabb[3:]))
            pts, dirs, _, _, mask = lattice(sb, surrounding_point_step, surrounding_angle_step)
            pts = pts[~mask]
            dirs = draw(dirs, pts.shape[0])
            file("surrounding", pts, dirs, *self._teacher_query(pts, dirs))
        if global_point_step > 0:
            pts, dirs, _, _, mask = lattice(self.model.aabb_train.view(2, 3).cpu(), global_point_step, global_angle_step)
            pts = pts[~mask]
            dirs = draw(dirs, pts.shape[0])
            file("global", pts, dirs, *self._teacher_query(pts, dirs))
        self.invalidate_graphs()  # (the per-chunk graphs hold raw pointers into the previous point / target tensors)
        return n_local

    def invalidate_graphs(self):
        """Drop every captured graph that bakes in state this trainer can replace: the per-chunk pretraining graphs (slices
        of `pretraining_data`, the optimizer's moment tensors, the learning rate), and the teacher's proxy-render graph (the
        mapper's parameters are kernel arguments, the teacher's bitfield / tables are raw pointers).  Called by
        init_pretraining(), load_checkpoint() and set_teacher(); call it after changing the teacher's mapper or occupancy
        state in place."""
        self._pt_graphs, self._pt_padded = {}, {}
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
        """SealNeRF/trainer.py:491-503: set every group's lr; the lr of the FIRST group at that moment is remembered and a
        negative `lr` restores it (once).  As in the reference, a second set_lr(x) before the restore remembers x, not the
        original value: after two pretraining epochs `set_lr(-1)` leaves the pretraining lr in place until the scheduler's
        next step recomputes the groups from their initial lrs."""
        if lr < 0:
            if getattr(self, "_cached_lr", None) is None:
                return
            lr, self._cached_lr = self._cached_lr, None
        else:
            self._cached_lr = self.optimizer.param_groups[0]["lr"]
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        # an optimizer whose step has been captured once applies lr changes through a device-side factor (nerf/optim.py:
        # capture_lr): bring it up to date HERE, so that a graph replayed next — a pretraining chunk, the fine-tuning step —
        # runs at this lr and not at the factor some other phase left behind
        follow = getattr(self.optimizer, "follow_lr_schedule", None)
        if follow is not None and getattr(self.optimizer, "_lr_captured", None) is not None \
                and not torch.cuda.is_current_stream_capturing() and not follow():
            self.optimizer.capture_lr()   # groups moved by different factors: rebase = a new lr_epoch, every graph re-captures

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

    def _pretrain_chunk(self, part, k, sl, n_total):
        """one optimizer step on the (static) chunk `sl` of part `part`; GPU + native optimizer: captured once per chunk
        and replayed — the chunk's tensors never move, the learning rate and the frozen MLPs are part of the capture"""
        src = self.pretraining_data[part]
        key = (part, k)
        args = (src["points"][sl], src["dirs"][sl], src["sigma"][sl], src["color"][sl])
        on_gpu = args[0].is_cuda
        if on_gpu and args[0].shape[0] % 128:
            # the chunk's points and directions padded to whole 128-row tiles ONCE (they are static): pretrain_loss would pad
            # them again on every step otherwise (two fills + two copies per replay)
            cache = self._pt_padded
            ent = cache.get(key)
            if ent is None or ent[0] != (args[0].data_ptr(), args[0].shape[0]):
                pad = 128 - args[0].shape[0] % 128
                ent = ((args[0].data_ptr(), args[0].shape[0]), F.pad(args[0], (0, 0, 0, pad)), F.pad(args[1], (0, 0, 0, pad), value=1.0))
                cache[key] = ent
            args = (ent[1], ent[2], args[2], args[3])
        if not (self.graph_pretraining and on_gpu and self.native_optim):
            return self.pretrain_step(*args, n_total=n_total)
        # the entry is valid for exactly the tensors it was captured on (a replaced chunk tensor or optimizer moment = re-capture)
        # and for the lr base it baked in (nerf/optim.py: every capture_lr() — this trainer's or the fine-tuning graph's — is a
        # new lr_epoch; an entry of an older epoch would run at old_base x new_factor)
        sig = (key, args[0].data_ptr(), args[2].data_ptr(), tuple(st["exp_avg"].data_ptr() for st in self.optimizer.state.values()
                                                                   if "exp_avg" in st), float(self.pretraining_lr),
               getattr(self.optimizer, "lr_epoch", 0))
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
            sig = sig[:-1] + (getattr(self.optimizer, "lr_epoch", 0),)  # (the eager step may have rebased: store what the capture sees)
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
        """one pass over every part of the pretraining set (SealNeRF/trainer.py:363-452: `set_lr(pretraining_lr)`, force-fill
        the bitfield, training mode, frozen MLPs, then local -> surrounding -> global in steps of the batch size); every rank
        processes its shard of each chunk.  As in the reference the epoch leaves the MLPs frozen and the pretraining lr set:
        `end_pretraining()` — called by the next fine-tuning step, as `train()` does before every training epoch (:338-339) —
        unfreezes and restores.  Returns the mean of the step losses."""
        from parallel import shard_slice
        self.set_lr(self.pretraining_lr)
        if not self.model.density_bitfield_hacked:
            self.model.hack_bitfield()
        self.model.train()
        self.freeze_mlp(True)
        self._pretraining_open = True
        rank, world = (self.dist.rank, self.dist.world) if self.dist is not None else (0, 1)
        losses = []
        for part, src in self.pretraining_data.items():
            for k, (a, b) in enumerate(zip(src["steps"][:-1], src["steps"][1:])):
                if b <= a:
                    continue
                lo, hi = shard_slice(b - a, rank, world)
                losses.append(self._pretrain_chunk(part, k, slice(a + lo, a + hi), b - a))
        self.last_pretrain_losses = losses
        if not losses:
            return torch.zeros(())
        return losses[0] if len(losses) == 1 else torch.stack([l.reshape(()) for l in losses]).mean()

    def end_pretraining(self):
        """`self.freeze_mlp(False); self.set_lr(-1)` (SealNeRF/trainer.py:338-339)"""
        if getattr(self, "_pretraining_open", False):
            self._pretraining_open = False
            self.freeze_mlp(False)
            self.set_lr(-1)

    # ------------------------------------------------------------------ global fine-tuning
    online_proxy_mode = None  # None: the teacher renders in the mode it is in (the reference); "train" / "eval": forced

    @torch.no_grad()
    def proxy_truth(self, rays_o, rays_d, out_rgb=None, out_depth=None, n_batch=1, teacher_mode=None):
        """teacher-rendered RGB + depth targets for a ray batch — the render call of SealNeRF/trainer.py:506-586
        (`teacher_model.render(..., staged=True, bg_color=None, perturb=False, force_all_rays=True)`, both outputs through
        nan_to_num).  `out_rgb` [N,3] / `out_depth` [N]: write them there (fp32, contiguous).

        The teacher renders in the mode it is in, as in the reference: main_SealNeRF.py:210 leaves it in eval mode, so the
        targets come from run_cuda's inference loop (depth measured from the camera, raymarching.cu:844-872).  In training
        mode the render takes the training branch — one march over every ray (`force_all_rays` is an argument of that branch
        only), depth measured from the first sample like the student's (:536-552).  `teacher_mode` ("train" | "eval") forces
        one of the two for this call.  `n_batch` > 1 renders the rays in pieces (:551-563, `proxy_batch`)."""
        if not self.teacher.density_bitfield_hacked:
            self.teacher.hack_bitfield()
        mode = teacher_mode or self.online_proxy_mode
        was_training = self.teacher.training
        if mode is not None:
            self.teacher.train(mode == "train")
        total = rays_o.shape[-2]
        if n_batch > 1 and rays_o.ndim == 3:
            size = total // n_batch
            pieces = n_batch + (1 if total % n_batch else 0)
            parts = [self.proxy_truth(rays_o[:, i * size:(i + 1) * size], rays_d[:, i * size:(i + 1) * size], teacher_mode=mode)
                     for i in range(pieces)]
            self.teacher.train(was_training)
            rgb, dep = torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1)
            if out_rgb is not None:
                torch._foreach_copy_([out_rgb, out_depth], [rgb.reshape(-1, 3), dep.reshape(-1)])
                return out_rgb.view(rgb.shape), out_depth.view(dep.shape)
            return rgb, dep
        native = rays_o.is_cuda and (out_rgb is not None or self.native_optim) and self.teacher.training
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

    def init_proxy_cache(self, n_poses, n_pixels):
        """`cache_gt` (SealNeRF/trainer.py:300-311): per (pose, pixel) memo of the proxied targets"""
        dev = next(self.model.parameters()).device
        self.cache_gt = True
        self.proxy_cache_mask = torch.zeros(n_poses, n_pixels, dtype=torch.bool, device=dev)
        self.proxy_cache_image = torch.zeros(n_poses, n_pixels, 3, dtype=torch.float, device=dev)
        self.proxy_cache_depth = torch.zeros(n_poses, n_pixels, dtype=torch.float, device=dev)

    @torch.no_grad()
    def proxy_truth_data(self, data, all_ray=True, use_cache=False, n_batch=1):
        """SealNeRF/trainer.py:506-586 on a data-loader dict, in place: `images` / `depths` become the teacher's targets.
        `skip_proxy` (the provider already proxied its dataset, SealNeRF/provider.py:101): nothing happens.  A full frame
        (`images` or `images_shape` of 4 dims) comes back as [B, H, W, C].  With the pixel cache only the rays whose
        (data_index, pixel_index) entry is still empty are rendered."""
        if data.get("skip_proxy"):
            return
        is_full, shape = False, None
        if "images" in data:
            shape = data["images"].shape
            is_full = data["images"].ndim == 4
        elif "images_shape" in data:
            shape = data["images_shape"]
            is_full = len(shape) == 4
        use_cache = bool(use_cache and data.get("pixel_index") is not None and not is_full)
        rays_o, rays_d = data["rays_o"], data["rays_d"]
        if use_cache:
            di, pi = data["data_index"], data["pixel_index"]
            di = di.to(self.proxy_cache_mask.device) if torch.is_tensor(di) else di
            todo = ~self.proxy_cache_mask[di, pi]
            if todo.any():
                rgb, dep = self.proxy_truth(rays_o[todo][None], rays_d[todo][None], n_batch=n_batch)
                self.proxy_cache_image[di, pi[todo]] = rgb
                self.proxy_cache_depth[di, pi[todo]] = dep
                self.proxy_cache_mask[di, pi[todo]] = True
            data["images"], data["depths"] = self.proxy_cache_image[di, pi], self.proxy_cache_depth[di, pi]
        else:
            data["images"], data["depths"] = self.proxy_truth(rays_o, rays_d, n_batch=n_batch)
        if is_full:
            data["images"] = data["images"].reshape(*shape[:-1], -1)
            data["depths"] = data["depths"].reshape(*shape[:-1], -1)

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
        self.end_pretraining()
        if gt_rgb is None:
            gt_rgb, gt_depth = self.proxy_truth(rays_o, rays_d)
        self.model.train()
        self._maybe_update_extra_state()  # (with data parallelism: occupancy state re-synchronised over the ranks)
        self.global_step += 1
        loss = self._seal_step(rays_o, rays_d, gt_rgb, gt_depth, bg_color)
        self._sched_step()
        return loss


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
            self.proxy_truth(self.s_ro, self.s_rd, self.s_gt, self.s_depth, teacher_mode="train")
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
        self.end_pretraining()
        if gt_rgb is None:
            mode = self.online_proxy_mode
            ok = (self.graph_proxy and self.fp16 and rays_o.is_cuda and rays_o.numel() == self.s_ro.numel()
                  and (mode == "train" or (mode is None and self.teacher.training))  # (the one-march branch has static extents)
                  and self.teacher.honours_row_limit_under_autocast(rays_o.numel() // 3 * self.render_kwargs["max_steps"]))
            if ok:
                torch._foreach_copy_([self.s_ro, self.s_rd], [rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)])
                self._proxy_replay()
                gt_rgb, gt_depth = self.s_gt, self.s_depth
            else:
                gt_rgb, gt_depth = self.proxy_truth(rays_o, rays_d)
        return GraphedTrainer.train_step(self, rays_o, rays_d, (gt_rgb, gt_depth), bg_color)
