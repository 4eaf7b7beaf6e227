"""gridencoder — drop-in for the reference's `gridencoder` package on MI355X.

Same `GridEncoder` constructor / attributes / forward and `grid_encode`
Function surface as gridencoder/grid.py of the reference; native work goes to
libseal3d_hip (s3d_hip.GridBackend).
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook

import s3d_hip

_backend = s3d_hip.GridBackend

_GRIDTYPE = {"hash": 0, "tiled": 1}
_INTERP = {"linear": 0, "smoothstep": 1}


_weights_epoch = 0


def bump_weights_epoch():
    """Called by trainers that update parameters WITHOUT Python-side in-place ops (HIP-graph replays do not bump
    `Tensor._version`), so that cached fp16 tables are refreshed.  Anything that writes parameters through `.data`
    (`reset_parameters`, an EMA's `copy_to` / `restore`) has to call it too: `.data` writes bump no version either."""
    global _weights_epoch
    _weights_epoch += 1


def _weights_epoch_now():
    return _weights_epoch


def _on_optimizer_step(optimizer, args, kwargs):
    bump_weights_epoch()


# torch's fused optimizers (Adam(fused=True) under GradScaler is the reference trainer's default on GPU) update the
# parameters in place WITHOUT bumping `Tensor._version` (measured: version stays 0 across steps), so the version alone
# cannot key a cache of casts: every torch optimizer step, of any optimizer, advances the epoch.
_register_step_hook(_on_optimizer_step)


def _half_table(embeddings, cache):
    """fp16 copy of the table for autocast (grid.py:41-44 casts on every call).  In eval mode (the inference loop calls
    the encoder ~50 times per frame on unchanged weights) the copy is cached per parameter version: one 73 MB cast per
    weight update instead of one per call."""
    if not cache or torch.cuda.is_current_stream_capturing():
        return embeddings.to(torch.half)
    # the copy lives ON the parameter object, so it is freed with it (a replaced teacher's 73 MB table does not stay pinned
    # in a module-level dict)
    hit = getattr(embeddings, "_s3d_eval_half", None)
    if hit is not None and hit[0] == embeddings._version and hit[1] == _weights_epoch and hit[2].device == embeddings.device:
        return hit[2]
    h = embeddings.detach().to(torch.half)
    embeddings._s3d_eval_half = (embeddings._version, _weights_epoch, h)
    return h


class _GridEncode(Function):
    """grid.py:24-89.  Autocast is handled by hand: inputs stay fp32, the table is cast to half when C is even."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                gridtype=0, align_corners=False, interpolation=0, cache_half=False, level_major=False, bound=0.0,
                n_valid=None, live=None):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = base_resolution

        param = embeddings
        if torch.is_autocast_enabled("cuda") and C % 2 == 0:
            half = getattr(param, "_s3d_half", None)  # fp16 copy maintained by nerf.optim.NativeAdam
            if half is not None and param._s3d_half_version == param._version:
                embeddings = half
            else:
                embeddings = _half_table(embeddings, cache_half)
        embeddings = embeddings.contiguous()

        outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)  # level-major
        dy_dx = (torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype)
                 if calc_grad_inputs else None)
        # (keywords only when used: a backend without the fused normalisation / padded batches is never asked for them)
        extra = {}
        if bound:
            extra["bound"] = bound
        if n_valid is not None:
            extra["n_valid"] = n_valid  # device sample count of a padded batch: rows beyond it are not produced
        fwd_extra = dict(extra)
        if live is not None and not torch.is_grad_enabled():
            fwd_extra["live"] = live  # inference: rows marked dead are written as zeros without reading the table
        _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype,
                                     align_corners, interpolation, **fwd_extra)
        if not level_major:  # (a fused consumer reads the kernel's own [L, B, C] layout in place: ffmlp input_layout=1)
            outputs = outputs.permute(1, 0, 2).reshape(B, L * C)

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.meta = (B, D, C, L, S, H, gridtype, interpolation, align_corners, level_major, bound)
        ctx.param = param
        ctx.extra = extra
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation, align_corners, level_major, bound = ctx.meta
        if level_major:
            grad = grad.contiguous()  # already [L, B, C]
        else:
            grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()  # [L, B, C]
        # fp16 hand-over (nerf.optim.NativeAdam): the table gradient is ACCUMULATED into the optimizer's fp16 buffer and
        # not returned to autograd, which would cast it to fp32 and add it into `.grad` (220 MB of traffic per step)
        stash = getattr(ctx.param, "_s3d_grad", None)
        if stash is not None and stash.dtype == embeddings.dtype and stash.shape == embeddings.shape:
            grad_embeddings = stash
            was_touched = getattr(ctx.param, "_s3d_grad_touched", False)
            ctx.param._s3d_grad_touched = True
            found_inf = getattr(ctx.param, "_s3d_found_inf", None)  # GradScaler's check made by the writing kernel
        else:
            stash = None
            grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        extra = ctx.extra
        if stash is not None and found_inf is not None:
            extra = dict(extra, found_inf=found_inf)
        # the optimizer armed this table for THIS backward pass (nerf/optim.py: NativeAdam.arm_fused_tables): the accumulate
        # kernel applies the update itself and no gradient is written
        armed = ctx.param.__dict__.pop("_s3d_fused_arm", None) if stash is not None else None
        if armed is not None and dy_dx is None and hasattr(_backend, "grid_encode_backward_adam"):
            if _backend.grid_encode_backward_adam(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, gridtype,
                                                  align_corners, interpolation, armed, **extra):
                ctx.param._s3d_grad_touched = was_touched  # (nothing was written into the hand-over buffer)
                ctx.param._s3d_fused_done = True
            return None, None, None, None, None, None, None, None, None, None, None, None, None, None
        _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx,
                                      grad_inputs, gridtype, align_corners, interpolation, **extra)
        if grad_inputs is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, (None if stash is not None else grad_embeddings), None, None, None, None, None, None, None, None, None, None, None, None


grid_encode = _GridEncode.apply


def _table_for_kernels(param, cache_half):
    """(table the kernels read, Parameter): the optimizer's fp16 copy, a cached cast, or the parameter itself (_GridEncode.forward)"""
    emb = param
    if torch.is_autocast_enabled("cuda") and param.shape[1] % 2 == 0:
        half = getattr(param, "_s3d_half", None)
        if half is not None and param._s3d_half_version == param._version:
            emb = half
        else:
            emb = _half_table(param, cache_half)
    return emb.contiguous()


class _GridEncodePair(Function):
    """Two encoders of identical geometry on the same points (the density and the colour encoder of nerf/network.py:99-128) in ONE
    forward launch (s3d_grid_encode_forward_pair); level-major outputs, no input gradient; backward = the two encoders' backward
    calls, as two _GridEncode nodes would make them."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, emb_a, emb_b, offsets, per_level_scale, base_resolution, gridtype, align_corners, interpolation,
                cache_half, bound, n_valid, live):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = emb_a.shape[1]
        S = float(np.log2(per_level_scale))
        H = base_resolution
        pa, pb = emb_a, emb_b
        ta, tb = _table_for_kernels(pa, cache_half), _table_for_kernels(pb, cache_half)
        out_a = torch.empty(L, B, C, device=inputs.device, dtype=ta.dtype)
        out_b = torch.empty(L, B, C, device=inputs.device, dtype=ta.dtype)
        extra = {}
        if bound:
            extra["bound"] = bound
        if n_valid is not None:
            extra["n_valid"] = n_valid
        fwd_extra = dict(extra)
        if live is not None and not torch.is_grad_enabled():
            fwd_extra["live"] = live
        _backend.grid_encode_forward_pair(inputs, ta, tb, offsets, out_a, out_b, B, D, C, L, S, H, gridtype, align_corners,
                                          interpolation, **fwd_extra)
        ctx.save_for_backward(inputs, ta, tb, offsets)
        ctx.meta = (B, D, C, L, S, H, gridtype, interpolation, align_corners)
        ctx.params = (pa, pb)
        ctx.extra = extra
        ctx.set_materialize_grads(False)
        return out_a, out_b

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, ga, gb):
        inputs, ta, tb, offsets = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation, align_corners = ctx.meta
        outs = []
        for k, (grad, table, param) in enumerate(((ga, ta, ctx.params[0]), (gb, tb, ctx.params[1]))):
            if grad is None or not ctx.needs_input_grad[1 + k]:  # (a frozen table of the pair: no scatter, no hand-over mark)
                outs.append(None)
                continue
            grad = grad.contiguous()
            stash = getattr(param, "_s3d_grad", None)
            extra = ctx.extra
            if stash is not None and stash.dtype == table.dtype and stash.shape == table.shape:
                grad_embeddings = stash
                param._s3d_fused_prev_touched = getattr(param, "_s3d_grad_touched", False)
                param._s3d_grad_touched = True
                found_inf = getattr(param, "_s3d_found_inf", None)
                if found_inf is not None:
                    extra = dict(extra, found_inf=found_inf)
            else:
                stash = None
                grad_embeddings = torch.zeros_like(table)
            # in-backward update (nerf/optim.py: arm_fused_tables) for the LAST table of the pair only: its kernel takes the step's
            # skip decision when every poison word of the pair is known — the first table's scatter has run by then, but not
            # vice versa — so the first table keeps the separate update
            armed = param.__dict__.pop("_s3d_fused_arm", None) if stash is not None else None
            if armed is not None and k == 1 and hasattr(_backend, "grid_encode_backward_adam"):
                was = getattr(param, "_s3d_fused_prev_touched", False)
                if _backend.grid_encode_backward_adam(grad, inputs, table, offsets, grad_embeddings, B, D, C, L, S, H, gridtype,
                                                      align_corners, interpolation, armed, **extra):
                    param._s3d_grad_touched = was
                    param._s3d_fused_done = True
                outs.append(None)
                continue
            _backend.grid_encode_backward(grad, inputs, table, offsets, grad_embeddings, B, D, C, L, S, H, None, None, gridtype,
                                          align_corners, interpolation, **extra)
            outs.append(None if stash is not None else grad_embeddings)
        return (None, outs[0], outs[1]) + (None,) * 10


def grid_encode_pair(enc_a, enc_b, inputs, bound=1, n_valid=None, live=None):
    """level-major outputs of two GridEncoder modules of identical geometry for the same [B, D] points from one launch, or None
    when the pair call does not apply (the caller then encodes one after the other)"""
    if not (getattr(_backend, "supports_bound", False) and hasattr(_backend, "grid_encode_forward_pair") and inputs.is_cuda
            and inputs.dim() == 2 and not inputs.requires_grad and inputs.dtype == torch.float32 and bound > 0
            and math.frexp(2.0 * bound)[0] == 0.5):
        return None
    same = (enc_a.embeddings.shape == enc_b.embeddings.shape and enc_a.per_level_scale == enc_b.per_level_scale
            and enc_a.base_resolution == enc_b.base_resolution and enc_a.gridtype_id == enc_b.gridtype_id
            and enc_a.align_corners == enc_b.align_corners and enc_a.interp_id == enc_b.interp_id
            and enc_a.input_dim == enc_b.input_dim and enc_a.training == enc_b.training
            # (the offsets follow from these and the table size — compared on the host: a tensor comparison would synchronise,
            #  which a stream that is being captured does not allow)
            and enc_a.num_levels == enc_b.num_levels and enc_a.level_dim == enc_b.level_dim
            and enc_a.max_params == enc_b.max_params)
    C = enc_a.embeddings.shape[1]
    halfed = torch.is_autocast_enabled("cuda") and C % 2 == 0
    if not same or (C * (2 if halfed else enc_a.embeddings.element_size())) % 4:  # (the lane-pair kernel: whole 32-bit feature words)
        return None
    return _GridEncodePair.apply(inputs, enc_a.embeddings, enc_b.embeddings, enc_a.offsets, enc_a.per_level_scale,
                                 enc_a.base_resolution, enc_a.gridtype_id, enc_a.align_corners, enc_a.interp_id,
                                 not enc_a.training, float(bound), n_valid, live)


class GridEncoder(nn.Module):
    """grid.py:96-185"""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:
            # geometric growth that lands on `desired_resolution` at the last level (grid.py:101-102)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))

        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _GRIDTYPE[gridtype]
        self.interpolation = interpolation
        self.interp_id = _INTERP[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        # per-level row counts: dense while (res+1)^D fits, capped at 2^log2_hashmap_size, padded to x8
        sizes = []
        for lvl in range(num_levels):
            res = int(np.ceil(base_resolution * per_level_scale ** lvl))
            rows = min(self.max_params, (res if align_corners else res + 1) ** input_dim)
            sizes.append(int(np.ceil(rows / 8) * 8))
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        total = int(offsets[-1])
        self.n_params = self.offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(total, level_dim))
        self.embeddings._s3d_stash_ok = True  # a native optimizer may take this table's gradient as the fp16 buffer
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)
        bump_weights_epoch()  # (a `.data` write bumps no version: cached fp16 copies are stale now)

    def __repr__(self):
        top = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {top} per_level_scale={self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} gridtype={self.gridtype} align_corners={self.align_corners} "
                f"interpolation={self.interpolation}")

    def forward(self, inputs, bound=1, level_major=False, n_valid=None, live=None):
        """`level_major=True` (build extension, 2-D inputs only) returns the kernel's own [num_levels, B, level_dim] layout
        instead of the reference's [B, num_levels * level_dim] — what ffmlp's `input_layout=1` consumes without a copy.
        `n_valid` (build extension): int32 GPU tensor, the sample count of a padded batch; rows past it (rounded up to
        128) are neither encoded nor back-propagated (seal3d_hip.h).  `live` (build extension, no-grad calls): fp32 GPU
        tensor with one row per input; rows whose first element is 0 are encoded as zeros without touching the table."""
        # [-bound, bound] -> [0, 1] (grid.py:146): inside the kernels when the backend can, no input gradient is needed and
        # 2 * bound is a power of two (the usual 1, 2, 4, ...: dividing and multiplying by the reciprocal then round
        # identically, so the fused result is bit-identical to the torch expression), else here
        fuse_bound = 0.0
        if getattr(_backend, "supports_bound", False) and inputs.is_cuda and not inputs.requires_grad \
                and inputs.dtype == torch.float32 and bound > 0 and math.frexp(2.0 * bound)[0] == 0.5:
            fuse_bound = float(bound)
        else:
            inputs = (inputs + bound) / (2 * bound)
        lead = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                          inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id,
                          not self.training, level_major, fuse_bound, n_valid, live)
        if level_major:
            return out
        return out.view(lead + [self.output_dim])

    @torch.autocast("cuda", enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000, grad_scale=None):
        """grid.py:162-185: adds the TV gradient into `embeddings.grad` (call between backward() and step()).
        With nerf.optim.NativeAdam the table gradient lives in the optimizer's fp16 buffer, multiplied by the loss scale:
        pass that scale as `grad_scale` (float or tensor, e.g. `scaler._scale`); the TV term is computed in fp32, scaled
        and added to the buffer."""
        D, C, L = self.input_dim, self.embeddings.shape[1], self.offsets.shape[0] - 1
        S, H = float(np.log2(self.per_level_scale)), self.base_resolution
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        stash = getattr(self.embeddings, "_s3d_grad", None)
        if stash is not None and getattr(self.embeddings, "_s3d_grad_touched", False):
            if grad_scale is None:
                raise ValueError("grad_total_variation: this table's gradient is the optimizer's loss-scaled fp16 buffer "
                                 "(nerf.optim.NativeAdam); pass grad_scale=<the scaler's scale>")
            tv = torch.zeros_like(self.embeddings, dtype=torch.float32)
            _backend.grad_total_variation(inputs.contiguous(), self.embeddings.detach().float(), tv, self.offsets, weight, B, D, C, L,
                                          S, H, self.gridtype_id, self.align_corners)
            stash.add_((tv * grad_scale).to(stash.dtype))
            self.embeddings._s3d_unchecked = True  # written by a torch op: the scaler checks the buffer itself
            return
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        _backend.grad_total_variation(inputs.contiguous(), self.embeddings, self.embeddings.grad, self.offsets, weight,
                                      B, D, C, L, S, H, self.gridtype_id, self.align_corners)
