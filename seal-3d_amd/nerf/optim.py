"""Parameter update of the training step, taken straight from the gradients the HIP backward kernels produce.

The reference's step is `torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15)` under `torch.cuda.amp.GradScaler`
(nerf/utils.py:356-361, 495-537; main_SealNeRF.py:283-288).  `NativeAdam` / `NativeGradScaler` are the same update
rule and the same loss-scale schedule, re-plumbed for a 12.2 M-row hash table whose gradient is fp16:

* a `GridEncoder` whose parameter has been adopted by `NativeAdam` hands its table gradient over as the fp16 buffer
  the backward kernel wrote (`param._s3d_grad`), instead of returning it to autograd (which would cast it to fp32,
  73 MB, and accumulate it into `.grad`, 147 MB);
* `NativeGradScaler.step` = one read of every gradient for the non-finite check, then `s3d_adam_step` per tensor —
  unscale, Adam and the fp16 copy of the updated table for the next autocast forward (`param._s3d_half`) in one pass;
  all of it skipped when a gradient is non-finite, exactly like `GradScaler.step`;
* everything stays on the device (step count, scale, found_inf), so the step is HIP-graph capturable.

Both classes exist only for GPU training with libseal3d_hip; the torch optimizer + GradScaler path of the trainers is
unchanged and remains the reference behaviour (CPU tests, A/B runs).
"""
import torch

import s3d_hip

_backend = s3d_hip.OptimBackend


class NativeAdam(torch.optim.Optimizer):
    """Adam (no amsgrad, no weight decay) with device-side step count.  `adopt_half_grads=True` switches every
    GridEncoder-style parameter that carries `_s3d_stash_ok` to the fp16 hand-over described in the module docstring."""

    def __init__(self, params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, adopt_half_grads=True, consume_grads=True):
        # (the torch.optim.Adam keys this class has no use for keep `state_dict()` loadable by torch.optim.Adam)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self.step_count = None
        # device-side factor on every group's lr (None: none).  A trainer that replays the step from a HIP graph keeps the
        # group's `lr` of the capture in the launch arguments and writes schedule / captured here (follow_lr_schedule)
        self.lr_scale = None
        self._lr_captured = None
        self.lr_epoch = 0  # bumped by every capture_lr(): a graph stores the epoch it baked its lr arguments in
        # the update launch clears every handed-over gradient behind its read: the next zero_grad() has nothing to fill
        self.consume_grads = consume_grads
        self.flat_half = None  # ONE fp16 buffer behind every handed-over gradient: one clear, one check, one all-reduce
        adopted, packs = [], []
        for group in self.param_groups:
            for p in group["params"]:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("NativeAdam updates fp32 parameters on the GPU")
                st = self.state[p]
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if self.step_count is None:
                    self.step_count = torch.zeros(1, dtype=torch.float32, device=p.device)
                if adopt_half_grads and getattr(p, "_s3d_stash_ok", False):
                    adopted.append(p)
                pk = getattr(p, "_s3d_pack_spec", None) if adopt_half_grads else None
                if pk is not None and pk[0].usable_on(p.device) and not any(pk[0] is q for q in packs):
                    packs.append(pk[0])
        # PackedWeights (nerf/network.py): nn.Linear weights that the fused MLP kernels read from ONE padded fp16 buffer per
        # network.  The optimizer takes their gradient where the MLP backward writes it (the pack's fp16 gradient twin, a region
        # of the flat buffer) and writes the updated fp16 weights straight into the pack — no cat / pad / cast per step, no
        # split of the fp16 weight gradient back into five fp32 `.grad`s.
        packs = [pk for pk in packs if all(id(m) in {id(q) for g in self.param_groups for q in g["params"]} for m, _, _ in pk.members)]
        if adopted or packs:
            sizes = [(p.numel() + 7) // 8 * 8 for p in adopted]  # 16-byte aligned views
            psizes = [(pk.numel + 7) // 8 * 8 for pk in packs]
            dev = (adopted[0] if adopted else packs[0].members[0][0]).device
            self.flat_half = torch.zeros(sum(sizes) + sum(psizes), dtype=torch.float16, device=dev)
            off = 0
            self.flat_half._s3d_param_cuts = []  # parameter boundaries (elements): where a chunked all-reduce may cut
            for p, n in zip(adopted, sizes):
                self.flat_half._s3d_param_cuts.append(off)
                p._s3d_flat_range = (off, off + p.numel())
                p._s3d_grad = self.flat_half[off:off + p.numel()].view(p.shape)
                p._s3d_grad_flat = self.flat_half
                p._s3d_grad_touched = False
                p._s3d_grad_consumed = False
                p._s3d_half = p.detach().to(torch.float16)
                p._s3d_half_version = p._version
                off += n
            for pk, n in zip(packs, psizes):
                self.flat_half._s3d_param_cuts.append(off)
                pk.adopt(self.flat_half[off:off + pk.numel], self.flat_half, (off, off + pk.numel))
                off += n
        self.packs = packs

    fuse_table_updates = True  # single replica: a hash table's update rides in its backward's accumulate kernel (arm_fused_tables)

    def arm_fused_tables(self, grad_scale=None):
        """Called by a trainer right before the ONE backward pass of a step it will follow with `step()` (single replica, every
        other gradient of the step checked where it is produced): each adopted fp16 C = 2 table is armed — its backward
        (gridencoder/grid.py) hands this state to s3d_grid_encode_backward_adam, the accumulate kernel applies the update where
        the row sums are, and `step()` leaves the table alone.  A table whose backward does not run, or falls back to the plain
        kernels, is updated by `step()` as always.  Returns the number of tables armed."""
        n = 0
        if not self.fuse_table_updates or self.flat_half is None:
            return 0
        lr_of = {}
        if self._lr_captured is not None:
            lr_of = {id(g): lr for g, lr in zip(self.param_groups, self._lr_captured)}
        for group in self.param_groups:
            for p in group["params"]:
                half = getattr(p, "_s3d_half", None)
                if (getattr(p, "_s3d_grad", None) is None or getattr(p, "_s3d_pack_spec", None) is not None or not p.requires_grad
                        or p.dim() != 2 or p.shape[1] != 2 or half is None or p._s3d_half_version != p._version
                        or getattr(p, "_s3d_found_inf", None) is None or float(p.__dict__.get("_s3d_l1", 0.0)) != 0.0):
                    continue
                st = self.state[p]
                b1, b2 = group["betas"]
                p._s3d_fused_arm = dict(param=p.data, exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"], param_half=half,
                                        lr=lr_of.get(id(group), group["lr"]), betas=(b1, b2), eps=group["eps"], step=self.step_count,
                                        grad_scale=grad_scale, lr_scale=self.lr_scale)
                n += 1
        return n

    def disarm_fused_tables(self):
        for group in self.param_groups:
            for p in group["params"]:
                p.__dict__.pop("_s3d_fused_arm", None)

    def grads(self):
        """(param, gradient tensor) for every parameter that has one: the fp16 hand-over buffer or `.grad`"""
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_s3d_fused_done", False):
                    continue  # (updated inside its backward's accumulate kernel this step)
                g = getattr(p, "_s3d_grad", None)
                if g is None or not getattr(p, "_s3d_grad_touched", False):
                    g = p.grad  # (a table no backward pass touched this step has no gradient, like `.grad is None`)
                if g is not None:
                    yield group, p, g

    def zero_grad(self, set_to_none=True):
        if self.flat_half is not None:
            # the backward kernels ACCUMULATE into it (several calls per step are summed).  Written since the last clear and not
            # consumed by step(): fill; otherwise it is still all zeros
            dirty = any(getattr(p, "_s3d_grad_touched", False) and not getattr(p, "_s3d_grad_consumed", False)
                        for group in self.param_groups for p in group["params"] if getattr(p, "_s3d_grad", None) is not None)
            if dirty:
                self.flat_half.zero_()
            # (under graph capture this decision is frozen into the graph: a capture that finds the buffer clean records no
            #  fill and relies on every replay being preceded by a consuming step — GraphedTrainer._replay checks the flags
            #  again on the host before each replay, clear_unconsumed())
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_s3d_grad", None) is not None:
                    p._s3d_grad_touched = False
                    p._s3d_grad_consumed = False
                    p._s3d_unchecked = False
                p.__dict__.pop("_s3d_grad_checked", None)  # (a producer's "checked at the source" mark never outlives its step)
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def clear_unconsumed(self):
        """Gradients written into the hand-over buffer and NOT consumed by a step (an eager backward whose step never came):
        clear them now.  Called before a captured step is replayed — that step was recorded on a clean buffer, so it holds no
        fill of its own (the consuming Adam of the previous replay clears behind its read).  Host-side flags only."""
        if self.flat_half is None:
            return False
        dirty = False
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_s3d_grad", None) is not None and getattr(p, "_s3d_grad_touched", False) \
                        and not getattr(p, "_s3d_grad_consumed", False):
                    dirty = True
                    p._s3d_grad_touched = False
        if dirty:
            self.flat_half.zero_()
        return dirty

    def resync_half(self):
        """re-make the fp16 copies after the fp32 parameters were written from outside (checkpoint load)"""
        for group in self.param_groups:
            for p in group["params"]:
                if hasattr(p, "_s3d_half"):
                    p._s3d_half.copy_(p.detach())
                    p._s3d_half_version = p._version

    def state_dict(self):
        """torch.optim.Adam's layout: per-parameter `step` next to exp_avg / exp_avg_sq (one shared count here)"""
        sd = super().state_dict()
        step = None if self.step_count is None else self.step_count.detach().reshape(()).cpu().clone()
        sd["state"] = {k: dict(st, step=step.clone()) for k, st in sd["state"].items()}  # (never the live dicts)
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = []
        for p, st in self.state.items():
            if "step" in st:
                steps.append(float(st.pop("step")))
            for k in ("exp_avg", "exp_avg_sq"):  # (torch casts the moments to the parameter's dtype/device already)
                st[k] = st[k].contiguous()
        if steps:
            self.step_count.fill_(max(steps))
        for group in self.param_groups:  # a parameter the checkpoint did not know starts from zero moments
            for p in group["params"]:
                st = self.state[p]
                for k in ("exp_avg", "exp_avg_sq"):
                    if k not in st:
                        st[k] = torch.zeros_like(p, memory_format=torch.contiguous_format)

    def capture_lr(self):
        """The step is about to be captured in a graph: remember each group's lr (it becomes a launch argument) and start
        the device-side factor at 1.  Every call is a new `lr_epoch`: a graph captured against an earlier base would run at
        old_base x new_factor, so whoever holds one compares the epoch it stored and re-captures (nerf/trainer.py,
        sealnerf/trainer.py)."""
        self.lr_epoch += 1
        if self.lr_scale is None:
            self.lr_scale = torch.ones(1, dtype=torch.float32, device=self.step_count.device)
        self._lr_captured = [float(g["lr"]) for g in self.param_groups]
        self._lr_factor = 1.0
        self.lr_scale.fill_(1.0)

    def follow_lr_schedule(self):
        """Before a replay: `param_groups[i]["lr"]` may have been moved by a scheduler (LambdaLR, main_SealNeRF.py:283-288)
        since the capture.  All groups moved by the same factor (the schedulers of the reference scale every group alike):
        that factor goes to the device word the captured Adam launch reads — one 4-byte fill, and only when it changed.
        Returns False when the groups moved by different factors: the caller re-captures."""
        if self._lr_captured is None:
            return True
        now = [float(g["lr"]) for g in self.param_groups]
        f = [n / c if c != 0 else (1.0 if n == 0 else float("inf")) for n, c in zip(now, self._lr_captured)]
        if any(abs(x - f[0]) > 1e-12 * max(1.0, abs(f[0])) for x in f) or f[0] == float("inf"):
            return False
        if f[0] != self._lr_factor:
            self.lr_scale.fill_(f[0])
            self._lr_factor = f[0]
        return True

    def mark_all_touched(self):
        """after a gradient all-reduce every stashed gradient is the replicas' mean on EVERY rank, also on a rank whose own
        batch never reached that table: all ranks must take the same update"""
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_s3d_grad", None) is not None and p.requires_grad:  # (frozen MLPs of Seal's pretraining stay put)
                    p._s3d_grad_touched = True

    @torch.no_grad()
    def step(self, grad_scale=None, found_inf=None, before_param=None, advance=True):
        """`before_param(p)`: called before parameter p is updated (data parallelism: wait for the all-reduce pieces that
        cover p's gradient while later pieces are still on the wire); without it all tensors are updated by ONE launch.
        `advance=False`: the caller advances `step_count` itself (NativeGradScaler.update folds it into its own launch)."""
        batch, stale, consumed = [], [], []
        lr_of = {}
        if self._lr_captured is not None:
            # launches carry the lr of the capture and the device-side factor carries the schedule — also for an eager step of
            # an optimizer whose step has been captured once (same arithmetic on both routes)
            if not torch.cuda.is_current_stream_capturing() and not self.follow_lr_schedule():
                self.capture_lr()  # groups moved apart: rebase — a new lr_epoch, so every trainer holding a graph re-captures
            lr_of = {id(g): lr for g, lr in zip(self.param_groups, self._lr_captured)}
        for group, p, g in self.grads():
            if before_param is not None:
                before_param(p)
            st = self.state[p]
            half = getattr(p, "_s3d_half", None)
            if half is not None and p._s3d_half_version != p._version:
                half = None  # somebody wrote the parameter through torch: the fp16 copy is re-made below
            b1, b2 = group["betas"]
            packed = getattr(p, "_s3d_pack_spec", None) is not None and g is getattr(p, "_s3d_grad", None)
            if half is not None and not packed and not half.is_contiguous():
                half = None  # a pack member updated from a plain `.grad` (nn.Linear route): its strided fp16 image is re-copied below
            item = (p.data, g, st["exp_avg"], st["exp_avg_sq"], half, lr_of.get(id(group), group["lr"]), b1, b2, group["eps"])
            # an L1 penalty whose gradient this update forms itself (tensoRF/utils.py announces it per step: read once, then cleared)
            l1 = float(p.__dict__.pop("_s3d_l1", 0.0))
            if l1 != 0.0 and (before_param is not None or packed):
                raise RuntimeError("NativeAdam: an in-update L1 term needs the one-launch route (no packed weights, no piecewise reduction)")
            if before_param is not None and packed:
                # (row-strided views of the pack: the multi-tensor entry point knows the layout.  The MLP backward OVERWRITES
                #  the pack's gradient twin, so nothing is cleared behind the read)
                _backend.adam_step_multi([item + (False,)], self.step_count, grad_scale, found_inf, lr_scale=self.lr_scale)
                p._s3d_grad_consumed = True
            elif before_param is not None:
                _backend.adam_step(*item, self.step_count, grad_scale, found_inf, self.lr_scale)
            elif packed:
                batch.append(item + (False,))
                consumed.append(p)  # (nothing to clear: the next backward overwrites the pack's gradient twin)
            else:
                # (a hand-over buffer is cleared behind the read; a `.grad` stays readable after the step)
                mine = self.consume_grads and g is getattr(p, "_s3d_grad", None)
                batch.append(item + (mine, l1))
                if mine:
                    consumed.append(p)
            if half is None and hasattr(p, "_s3d_half"):
                stale.append(p)
        if batch:
            _backend.adam_step_multi(batch, self.step_count, grad_scale, found_inf, lr_scale=self.lr_scale)
            for p in consumed:
                p._s3d_grad_consumed = True
        for p in stale:
            p._s3d_half.copy_(p.detach())  # (a strided view into the pack for packed parameters)
            p._s3d_half_version = p._version
        if advance:
            _backend.adam_advance(self.step_count, found_inf)
        for group in self.param_groups:  # (the marks of this step's in-backward updates; an arm nobody consumed goes with them)
            for p in group["params"]:
                p.__dict__.pop("_s3d_fused_done", None)
                p.__dict__.pop("_s3d_fused_arm", None)


class NativeGradScaler:
    """torch.amp.GradScaler's schedule (init 2^16, x2 after 2000 clean steps, x0.5 on overflow) driven by the flag the
    gradient check raises; `step()` does check + unscale + Adam, `update()` the schedule (aten::_amp_update_scale_)."""

    def __init__(self, device, enabled=True, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._scale = torch.full((1,), init_scale if enabled else 1.0, dtype=torch.float32, device=device)
        self._growth_tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self._found_inf = torch.zeros(1, dtype=torch.float32, device=device)
        self._one = torch.ones(1, dtype=torch.float32, device=device)  # unit inverse scale of the multi-tensor check
        self._advance = None  # step count of the optimizer whose advance rides in update()'s launch

    def scale(self, loss):
        return loss * self._scale.to(loss.dtype) if self.enabled else loss

    def get_scale(self):
        return float(self._scale.item())

    def backward(self, loss):
        """`scale(loss).backward()` without the multiply and the ones-like root: the scale IS the root gradient"""
        if self.enabled:
            loss.backward(gradient=self._scale.reshape(()).to(loss.dtype))
        else:
            loss.backward()

    def state_dict(self):
        """torch.amp.GradScaler.state_dict()'s keys (empty when disabled, like torch)"""
        if not self.enabled:
            return {}
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self._growth_tracker.item())}

    def load_state_dict(self, state):
        if not self.enabled or not state:
            return
        self._scale.fill_(state["scale"])
        self._growth_tracker.fill_(state["_growth_tracker"])
        self.growth_factor, self.backoff_factor = state["growth_factor"], state["backoff_factor"]
        self.growth_interval = state["growth_interval"]

    def attach(self, optimizer):
        """the kernels that WRITE the handed-over gradients (grid / ffmlp backward) raise this scaler's flag themselves
        (seal3d_hip.h: `found_inf` of s3d_grid_encode_backward / s3d_ffmlp_backward): no separate pass over the 24.5 MB
        buffer per step.  Single replica only: after an averaging all-reduce the REDUCED gradient is what must be checked."""
        for group in optimizer.param_groups:
            for p in group["params"]:
                if getattr(p, "_s3d_grad", None) is not None:
                    p._s3d_found_inf = self._found_inf
                    p._s3d_unchecked = False

    def _checked_at_source(self, optimizer):
        adopted = [p for group in optimizer.param_groups for p in group["params"] if getattr(p, "_s3d_grad", None) is not None]
        return bool(adopted) and all(getattr(p, "_s3d_found_inf", None) is self._found_inf and
                                     not getattr(p, "_s3d_unchecked", False) for p in adopted)

    def _check(self, optimizer):
        flat = getattr(optimizer, "flat_half", None)
        if flat is not None and not self._checked_at_source(optimizer):
            _backend.grads_nonfinite(flat, self._found_inf)  # every handed-over gradient in one pass
        self._check_plain(optimizer, flat)

    def _check_plain(self, optimizer, flat):
        """the gradients that are NOT hand-over buffers (`.grad` of the nn.Linear weights of the Seal net, ...): one
        multi-tensor launch for all of them (aten's check-and-unscale with a unit scale raises the same flag; one launch per
        tensor was 5 x 5 us of a 0.9 ms Seal step, profiles/r08_seal.md), the native per-tensor kernel for a single one"""
        # (a gradient whose producer raised THIS flag itself — tensoRF/network.py: the factor backward — is not read again;
        #  the mark is the producer's, per backward pass, and is consumed here)
        # (the mark is popped from EVERY parameter first: a mark whose backward was not followed by this scaler's step — a loss
        #  evaluated without a step, an exception, another scaler — must not excuse a later gradient made by another route)
        marks = {id(p): p.__dict__.pop("_s3d_grad_checked", None) for _, p, _ in optimizer.grads()}
        rest = [g for _, p, g in optimizer.grads() if (flat is None or g is not getattr(p, "_s3d_grad", None))
                and marks.get(id(p)) is not self._found_inf]
        if len(rest) > 1 and all(g.is_cuda and g.dtype == rest[0].dtype and g.layout == torch.strided for g in rest):
            torch._amp_foreach_non_finite_check_and_unscale_(rest, self._found_inf, self._one)
        else:
            for g in rest:
                _backend.grads_nonfinite(g, self._found_inf)

    def step(self, optimizer, dist=None):
        """check + unscale + Adam.  With a data-parallel layer (`dist`, parallel/dist.py) whose gradients have NOT been reduced
        yet, the reduction is pipelined with the update: the local gradients are checked first and the flag is reduced (MAX)
        next to the gradient pieces — RCCL's averaging cannot overflow, so "any rank saw a non-finite gradient" is exactly
        "the reduced gradient is non-finite" — and every parameter is updated as soon as the pieces covering its gradient
        have arrived, while the later pieces are still on the wire."""
        # (found_inf is cleared by update(); it starts at zero)
        self._check(optimizer)
        # the step-count advance of (one) optimizer is folded into update()'s launch: both are single-thread kernels
        fold = self.enabled and self._advance is None and getattr(optimizer, "step_count", None) is not None
        if fold:
            self._advance = optimizer.step_count
        if dist is None or (dist.world == 1 and not dist.force_collective):
            optimizer.step(grad_scale=self._scale if self.enabled else None, found_inf=self._found_inf, advance=not fold)
            return
        if not dist.fused_avg():
            # SUM + divide fallback (gloo, or no AVG): the sum of `world` finite loss-scaled fp16 gradients can overflow although
            # no rank saw a non-finite value, so it is the REDUCED buffers that are checked — all of them, before the first
            # parameter is touched (a step is skipped as a whole).  No pipelining on this path.
            dist.allreduce_grads()
            if hasattr(optimizer, "mark_all_touched"):
                optimizer.mark_all_touched()
            flat = getattr(optimizer, "flat_half", None)
            if flat is not None:
                _backend.grads_nonfinite(flat, self._found_inf)
            self._check_plain(optimizer, flat)
            dist.allreduce_flag(self._found_inf)  # (identical on every rank already; keeps the replicas' decisions tied)
            optimizer.step(grad_scale=self._scale if self.enabled else None, found_inf=self._found_inf, advance=not fold)
            return
        # the 4-byte flag goes first: a process group runs its collectives in issue order, and every update kernel reads the
        # flag — behind the gradient pieces it would hold the first Adam launch until the last piece has arrived
        dist.allreduce_flag(self._found_inf)
        pending = dist.allreduce_grads_async()
        if hasattr(optimizer, "mark_all_touched"):
            optimizer.mark_all_touched()

        def before_param(p):
            rng = getattr(p, "_s3d_flat_range", None)
            buf = getattr(p, "_s3d_grad_flat", None)
            while pending:
                h = pending[0]
                # pieces were issued in buffer order: everything up to the end of p's range (or, for a parameter of the
                # fp32 bucket, everything) must have arrived
                if rng is not None and h[0] is buf and h[1] >= rng[1]:
                    break
                dist.finish_chunk(pending.pop(0))
        optimizer.step(grad_scale=self._scale if self.enabled else None, found_inf=self._found_inf, before_param=before_param,
                       advance=not fold)
        while pending:
            dist.finish_chunk(pending.pop(0))

    def update(self, ring_push=None):
        """`ring_push` = (loss, counter, loss_ring, counter_ring, cursor): a graph-replayed trainer's end-of-step bookkeeping
        (OptimBackend.step_ring_push) rides in the same single-thread launch"""
        if self.enabled and ring_push is not None:
            _backend.step_epilogue(self._scale, self._growth_tracker, self._found_inf, self.growth_factor, self.backoff_factor,
                                   self.growth_interval, self._advance, *ring_push)
            self._advance = None
        elif self.enabled:
            _backend.scaler_update(self._scale, self._growth_tracker, self._found_inf, self.growth_factor,
                                   self.backoff_factor, self.growth_interval, self._advance)
            self._advance = None
        else:
            self._found_inf.zero_()
            if ring_push is not None:
                _backend.step_ring_push(*ring_push)
