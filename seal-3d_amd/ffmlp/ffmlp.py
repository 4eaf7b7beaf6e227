"""ffmlp — drop-in for the reference's `ffmlp` package (ffmlp/ffmlp.py) on MI355X.

`FFMLP(input_dim, output_dim, hidden_dim, num_layers, activation)` with one flat
fp32 `weights` parameter laid out [W,in] | (n-1)x[W,W] | [16,W] (row-major
[out,in] per layer), fp16 compute, MFMA kernels in libseal3d_hip.  The
`forward_buffer` / `backward_buffer` tensors keep the reference's shape
[num_layers, B, hidden] but their internal element order is private to the
library (MFMA fragment order), exactly as they are private scratch in the
reference.
"""
import math
import os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import s3d_hip

_backend = s3d_hip.FFMLPBackend

_FUSED_BACKWARD = os.environ.get("S3D_FFMLP_FUSED", "1") != "0"  # A/B switch: 0 = store activations, two-kernel backward

_ACTIVATIONS = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}


def convert_activation(act):
    """ffmlp.py:89-96 (anything unknown, incl. 'none', maps to 6 = identity)"""
    return _ACTIVATIONS.get(act, 6)


class _ParamRef:
    """Carries the fp32 Parameter through `custom_fwd(cast_inputs=...)` (which only touches tensors) to backward()."""
    __slots__ = ("param",)

    def __init__(self, param):
        self.param = param


class _FFMLPForward(Function):
    """ffmlp.py:15-83"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False, param_ref=None, hook=None, input_layout=0, n_valid=None,
                rgb_head=False):
        """`rgb_head` (build extension, seal3d_hip.h): the result is fp32 [B, 3] = sigmoid(output[:, :3]) straight from the last
        layer's store (the colour network's head), and backward() takes the gradient w.r.t. it"""
        B = inputs.shape[1] if input_layout else inputs.shape[0]
        # outside autocast `custom_fwd` does not cast: the kernels are fp16-only, so cast here (the autograd
        # engine converts the returned fp16 gradients back to the parameter dtype)
        inputs = inputs.to(torch.half).contiguous()
        weights = weights.to(torch.half).contiguous()
        rgb = torch.empty(B, 3, device=inputs.device, dtype=torch.float32) if rgb_head else None
        outputs = None if rgb_head else torch.empty(B, output_dim, device=inputs.device, dtype=inputs.dtype)
        if inference:
            # (the reference's inference_buffer is [B, hidden]; the layer-by-layer path of the widths the fused kernels do not
            #  cover ping-pongs between two such buffers, the fused kernels use none)
            native = getattr(_backend, "fused_backward_supported", None) is not None and \
                _backend.fused_backward_supported(input_dim, output_dim, hidden_dim, num_layers, activation)
            scratch = torch.empty((1,) if native else (2, B, hidden_dim), device=inputs.device, dtype=inputs.dtype)
            extra = {}
            if input_layout:
                extra["input_layout"] = input_layout
            if n_valid is not None:
                extra["n_valid"] = n_valid  # sync-free inference loop: rows behind the alive rays are skipped
            if rgb_head:
                extra["rgb_head"] = rgb
            _backend.ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                     output_activation, scratch, outputs, **extra)
            return rgb if rgb_head else outputs
        # The reference stores every layer's activations for the backward pass (forward_buffer [n, B, W]).  Where the
        # fused backward kernel covers the shape, nothing is stored: it re-computes the activations from `inputs` on chip
        # (256-384 B per sample less HBM traffic in each direction).
        fused = _FUSED_BACKWARD and getattr(_backend, "fused_backward_supported", None) is not None and \
            _backend.fused_backward_supported(input_dim, output_dim, hidden_dim, num_layers, activation)
        forward_buffer = None if fused else torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
        if input_layout and not fused:
            raise RuntimeError("FFMLP: the level-major input layout needs the fused backward kernel for this shape")
        if rgb_head and not fused:
            raise RuntimeError("FFMLP: the colour head needs the fused backward kernel for this shape")
        # (keywords only when used: a backend without these extensions is never asked for them)
        extra = {}
        if input_layout:
            extra["input_layout"] = input_layout
        if n_valid is not None:
            extra["n_valid"] = n_valid  # device sample count of a padded batch: rows beyond it are skipped
        _backend.ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                               output_activation, forward_buffer, outputs, **(dict(extra, rgb_head=rgb) if rgb_head else extra))
        ctx.rgb_head = rgb_head
        if rgb_head:
            ctx.save_for_backward(inputs, weights, rgb)
        elif fused:
            ctx.save_for_backward(inputs, weights)
        else:
            ctx.save_for_backward(inputs, weights, forward_buffer)
        ctx.meta = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs, fused)
        ctx.extra = extra
        ctx.param_ref = param_ref
        return rgb if rgb_head else outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        B = grad.shape[0]
        grad = grad.contiguous()
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs, fused = ctx.meta
        rgb = None
        if ctx.rgb_head:
            (inputs, weights, rgb), forward_buffer = ctx.saved_tensors, None
            grad = grad.float()
        elif fused:
            (inputs, weights), forward_buffer = ctx.saved_tensors, None
        else:
            inputs, weights, forward_buffer = ctx.saved_tensors
        # the reference zero-fills these three (ffmlp.py:67-73); the HIP kernels overwrite every element, so the
        # build allocates them uninitialised (saves two B x hidden x num_layers memsets per MLP per step)
        grad_inputs = (torch.empty_like(inputs) if calc_grad_inputs
                       else torch.zeros(1, device=grad.device, dtype=inputs.dtype))
        # fp16 hand-over (nerf.optim.NativeAdam): the reduce kernel ADDS the weight gradient to the optimizer's fp16 buffer
        # instead of returning it to autograd (fp32 cast + accumulate)
        stash = getattr(ctx.param_ref.param, "_s3d_grad", None) if ctx.param_ref is not None else None
        grad_weights = stash.view(weights.shape) if stash is not None else torch.empty_like(weights)  # every element is written
        backward_buffer = None if fused else torch.empty(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
        extra = dict(ctx.extra)
        if rgb is not None:
            extra["grad_rgb"], extra["rgb_head"], grad = grad, rgb, None
        if stash is not None:
            # (a packed nn.Linear pack, nerf/network.py: PackedWeights, is OVERWRITTEN: one backward per step writes it)
            extra["accumulate"] = not getattr(ctx.param_ref.param, "_s3d_overwrite", False)
            found_inf = getattr(ctx.param_ref.param, "_s3d_found_inf", None)  # GradScaler's check made by the writing kernel
            if found_inf is not None:
                extra["found_inf"] = found_inf
        _backend.ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim,
                                num_layers, activation, output_activation, calc_grad_inputs, backward_buffer,
                                grad_inputs, grad_weights, **extra)
        if stash is not None:
            ctx.param_ref.param._s3d_grad_touched = True
            grad_weights = None
        return ((grad_inputs if calc_grad_inputs else None), grad_weights) + (None,) * 13


ffmlp_forward = _FFMLPForward.apply


class _FFMLPNgpMid(Function):
    """The density network of nerf/network_ff.py with its head (seal3d_hip.h: mid_* of s3d_ffmlp_forward / _backward):
    (encoder output, dirs) -> sigma f32 [B] = trunc_exp(h[:, 0]), colour-net input f16 [B, 32] = [half(SH_4(d)) | h[:, 1:] | 0]
    from ONE launch per direction; the [B, 16] output h and its gradient never exist."""

    @staticmethod
    def forward(ctx, inputs, weights, dirs, dims, inference, param_ref, hook, input_layout, n_valid):
        input_dim, hidden_dim, num_layers, activation, output_activation = dims
        B = inputs.shape[1] if input_layout else inputs.shape[0]
        inputs = inputs.to(torch.half).contiguous()
        weights = weights.to(torch.half).contiguous()
        dirs = dirs.float().contiguous()
        sigma = torch.empty(B, device=inputs.device, dtype=torch.float32)
        cin = torch.empty(B, 32, device=inputs.device, dtype=torch.half)
        h0 = torch.empty(B, device=inputs.device, dtype=torch.half)
        extra = {}
        if input_layout:
            extra["input_layout"] = input_layout
        if n_valid is not None:
            extra["n_valid"] = n_valid
        fn = _backend.ffmlp_inference if inference else _backend.ffmlp_forward
        fn(inputs, weights, B, input_dim, 16, hidden_dim, num_layers, activation, output_activation, None, None,
           mid=(dirs, sigma, cin, h0), **extra)
        if not inference:
            ctx.save_for_backward(inputs, weights, h0)
            ctx.dims, ctx.extra, ctx.param_ref = dims, extra, param_ref
            ctx.calc_grad_inputs = ctx.needs_input_grad[0]  # (not `inputs.requires_grad`: a cast above would hide it)
            ctx.set_materialize_grads(False)
        return sigma, cin

    @staticmethod
    def backward(ctx, g_sigma, g_cin):
        inputs, weights, h0 = ctx.saved_tensors
        input_dim, hidden_dim, num_layers, activation, output_activation = ctx.dims
        B = h0.shape[0]
        if g_cin is None:
            g_cin = torch.zeros(B, 32, dtype=torch.half, device=h0.device)
        g_cin = g_cin.to(torch.half).contiguous()
        g_sigma = None if g_sigma is None else g_sigma.float().contiguous()
        calc = ctx.calc_grad_inputs
        grad_inputs = torch.empty_like(inputs) if calc else torch.zeros(1, device=h0.device, dtype=inputs.dtype)
        stash = getattr(ctx.param_ref.param, "_s3d_grad", None) if ctx.param_ref is not None else None
        grad_weights = stash.view(weights.shape) if stash is not None else torch.empty_like(weights)
        extra = dict(ctx.extra)
        if stash is not None:
            extra["accumulate"] = True
            found_inf = getattr(ctx.param_ref.param, "_s3d_found_inf", None)
            if found_inf is not None:
                extra["found_inf"] = found_inf
        _backend.ffmlp_backward(None, inputs, weights, None, B, input_dim, 16, hidden_dim, num_layers, activation,
                                output_activation, calc, None, grad_inputs, grad_weights, mid=(g_sigma, g_cin, h0), **extra)
        if stash is not None:
            ctx.param_ref.param._s3d_grad_touched = True
            grad_weights = None
        return ((grad_inputs if calc else None), grad_weights) + (None,) * 7


class _FFMLPNgpPair(Function):
    """Density network + head + colour network + sigmoid of nerf/network_ff.py:55-105 as ONE forward launch
    (s3d_ffmlp_ngp_pair_inference with its color_in / h0 outputs); backward = the colour network's fused backward (colour head)
    followed by the density network's (density head), exactly the two calls the separate Functions make."""

    @staticmethod
    def forward(ctx, inputs, w_sigma, w_color, dirs, dims_s, dims_c, refs, hook_s, hook_c, input_layout, n_valid):
        # (hook_s / hook_c: 0-dim leaves that keep the node in the graph when only the fp16 hand-over wants a gradient)
        B = inputs.shape[1] if input_layout else inputs.shape[0]
        inputs = inputs.to(torch.half).contiguous()
        w_sigma, w_color = w_sigma.to(torch.half).contiguous(), w_color.to(torch.half).contiguous()
        dirs = dirs.float().contiguous()
        sigma = torch.empty(B, device=inputs.device, dtype=torch.float32)
        rgb = torch.empty(B, 3, device=inputs.device, dtype=torch.float32)
        cin = torch.empty(B, 32, device=inputs.device, dtype=torch.half)
        h0 = torch.empty(B, device=inputs.device, dtype=torch.half)
        _backend.ngp_pair_inference(inputs, w_sigma, w_color, B, dims_s[1], dims_s[2], dims_c[2], dirs, sigma, rgb,
                                    input_layout, n_valid, cin, h0)
        ctx.save_for_backward(inputs, w_sigma, w_color, h0, cin, rgb)
        ctx.dims_s, ctx.dims_c, ctx.refs = dims_s, dims_c, refs
        ctx.extra = dict(({"input_layout": input_layout} if input_layout else {}), **({"n_valid": n_valid} if n_valid is not None else {}))
        ctx.calc_grad_inputs = ctx.needs_input_grad[0]  # (not `inputs.requires_grad`: a cast above would hide it)
        ctx.set_materialize_grads(False)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        inputs, w_sigma, w_color, h0, cin, rgb = ctx.saved_tensors
        B = h0.shape[0]
        dev = h0.device

        def target(ref, w):
            stash = getattr(ref.param, "_s3d_grad", None) if ref is not None else None
            extra = {}
            if stash is not None:
                extra["accumulate"] = not getattr(ref.param, "_s3d_overwrite", False)
                found_inf = getattr(ref.param, "_s3d_found_inf", None)
                if found_inf is not None:
                    extra["found_inf"] = found_inf
            return stash, (stash.view(w.shape) if stash is not None else torch.empty_like(w)), extra
        ref_s, ref_c = ctx.refs
        nv = {"n_valid": ctx.extra["n_valid"]} if "n_valid" in ctx.extra else {}
        # both calls leave their weight-gradient partial sums in their half of one scratch tensor: ONE reduce launch for the two
        in_c, W_c, nl_c, act_c, oact_c = ctx.dims_c
        in_s, W_s, nl_s, act_s, oact_s = ctx.dims_s
        nb_c = (_backend.backward_workspace_bytes(in_c, 16, W_c, nl_c) + 255) // 256 * 256
        nb_s = _backend.backward_workspace_bytes(in_s, 16, W_s, nl_s)
        scratch = torch.empty(nb_c + nb_s, dtype=torch.uint8, device=dev)
        ws_c, ws_s = scratch[:nb_c], scratch[nb_c:]
        # colour network: gradient w.r.t. its fp32 head output -> gradient of the colour-net input rows
        stash_c, gw_c, extra_c = target(ref_c, w_color)
        g_cin = torch.empty_like(cin)
        if g_rgb is None:
            g_rgb = torch.zeros_like(rgb)
        _backend.ffmlp_backward(None, cin, w_color, None, B, in_c, 16, W_c, nl_c, act_c, oact_c, True, None, g_cin, gw_c,
                                grad_rgb=g_rgb.float().contiguous(), rgb_head=rgb, workspace=ws_c, defer_reduce=True, **nv, **extra_c)
        # density network: the head's two gradients
        stash_s, gw_s, extra_s = target(ref_s, w_sigma)
        calc = ctx.calc_grad_inputs
        grad_inputs = torch.empty_like(inputs) if calc else torch.zeros(1, device=dev, dtype=inputs.dtype)
        g_sigma = None if g_sigma is None else g_sigma.float().contiguous()
        _backend.ffmlp_backward(None, inputs, w_sigma, None, B, in_s, 16, W_s, nl_s, act_s, oact_s, calc, None, grad_inputs, gw_s,
                                mid=(g_sigma, g_cin, h0), workspace=ws_s, defer_reduce=True, **ctx.extra, **extra_s)
        _backend.wgrad_reduce_pair((ws_c, B, in_c, W_c, nl_c, gw_c, extra_c.get("accumulate", False), extra_c.get("found_inf")),
                                   (ws_s, B, in_s, W_s, nl_s, gw_s, extra_s.get("accumulate", False), extra_s.get("found_inf")))
        for stash, ref in ((stash_s, ref_s), (stash_c, ref_c)):
            if stash is not None:
                ref.param._s3d_grad_touched = True
        return ((grad_inputs if calc else None), (None if stash_s is not None else gw_s), (None if stash_c is not None else gw_c)) + (None,) * 8


def _cached_half(w):
    """fp16 copy of the weights for inference calls (custom_fwd(cast_inputs=half) casts on EVERY call: two launches per
    iteration of the inference loop for a model that has no optimizer attached), cached per parameter version like the grid
    encoder's table (gridencoder/grid.py: _half_table)"""
    from gridencoder.grid import _half_table
    return _half_table(w, True)


class FFMLP(nn.Module):
    """ffmlp.py:99-169"""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16

        assert hidden_dim in [16, 32, 64, 128, 256], \
            f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"

        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.weights._s3d_stash_ok = True  # a native optimizer may take this gradient as an fp16 buffer (nerf/optim.py)
        self._autograd_hook = torch.zeros((), requires_grad=True)
        self.reset_parameters()
        _backend.allocate_splitk(self.num_layers + 1)

    def cleanup(self):
        _backend.free_splitk()

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)  # the reference reseeds the global RNG here (ffmlp.py:142)
        bound = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-bound, bound)
        from gridencoder.grid import bump_weights_epoch
        bump_weights_epoch()  # (a `.data` write bumps no version: cached fp16 copies are stale now)

    def forward(self, inputs):
        out = self.forward_padded(inputs)
        if self.padded_output_dim != self.output_dim:
            out = out[:, :self.output_dim]
        return out

    def fused_supported(self):
        """True when the hand-written MFMA kernels (re-computing fused backward, level-major input, `n_valid`, heads) serve
        this shape; False = the layer-by-layer path, which implements the reference's interface only"""
        return (_FUSED_BACKWARD and getattr(_backend, "fused_backward_supported", None) is not None
                and bool(_backend.fused_backward_supported(self.input_dim, self.padded_output_dim, self.hidden_dim,
                                                           self.num_layers, self.activation)))

    def rgb_head_supported(self):
        return (self.output_dim >= 3 and _FUSED_BACKWARD and getattr(_backend, "fused_backward_supported", None) is not None
                and _backend.fused_backward_supported(self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                                                      self.activation))

    def forward_ngp_mid(self, inputs, dirs, level_major=False, n_valid=None):
        """density network + head (see _FFMLPNgpMid): returns (sigma f32 [B], colour-net input f16 [B, 32])"""
        B = inputs.shape[1] if level_major else inputs.shape[0]
        if B % 128 != 0 or self.padded_output_dim != 16:
            raise RuntimeError("FFMLP.forward_ngp_mid: needs B % 128 == 0 and a 16-column output")
        w, ref, hook = self.weights, None, None
        if getattr(w, "_s3d_grad", None) is not None and getattr(w, "_s3d_half_version", None) == w._version:
            if torch.is_grad_enabled() and w.requires_grad:
                ref = _ParamRef(w)
                hook = self._autograd_hook
            w = w._s3d_half
        elif not torch.is_grad_enabled() and w.is_cuda:
            w = _cached_half(w)
        dims = (self.input_dim, self.hidden_dim, self.num_layers, self.activation, self.output_activation)
        return _FFMLPNgpMid.apply(inputs, w, dirs, dims, not self.training, ref, hook, 1 if level_major else 0, n_valid)

    def _inference_weights(self):
        """fp16 weights for a call that takes no gradient: the optimizer's copy while it is current, else one cast per version"""
        w = self.weights
        if getattr(w, "_s3d_grad", None) is not None and getattr(w, "_s3d_half_version", None) == w._version:
            return w._s3d_half
        return _cached_half(w) if w.is_cuda else w.half()

    def pair_supported(self, color_net):
        """can `forward_ngp_pair` serve this density network with `color_net`?  (the one-launch inference kernel: hidden 64,
        32-wide inputs, ReLU, 16-column outputs)"""
        return (self.hidden_dim == 64 and color_net.hidden_dim == 64 and self.input_dim == 32 and color_net.input_dim == 32
                and self.padded_output_dim == 16 and color_net.padded_output_dim == 16
                and self.activation == 0 and color_net.activation == 0 and self.output_activation == 6
                and color_net.output_activation == 6)

    def _train_weights(self):
        """(weights tensor for a call that records a gradient, _ParamRef for the fp16 hand-over, hook)"""
        w = self.weights
        if getattr(w, "_s3d_grad", None) is not None and getattr(w, "_s3d_half_version", None) == w._version:
            if torch.is_grad_enabled() and w.requires_grad:
                return w._s3d_half, _ParamRef(w), self._autograd_hook
            return w._s3d_half, None, None
        return w, None, None

    def forward_ngp_pair(self, inputs, dirs, color_net, level_major=False, n_valid=None):
        """density network + head + colour network + sigmoid in ONE launch (s3d_ffmlp_ngp_pair_inference): returns
        (sigma f32 [B], rgb f32 [B, 3]) — the same bits as forward_ngp_mid followed by color_net.forward_rgb; with gradients
        enabled the backward runs the two networks' fused backward kernels as the separate calls would"""
        B = inputs.shape[1] if level_major else inputs.shape[0]
        if B % 128 != 0 or not self.pair_supported(color_net):
            raise RuntimeError("FFMLP.forward_ngp_pair: needs B % 128 == 0 and two 64-wide ReLU networks with 32 inputs / 16 outputs")
        if torch.is_grad_enabled() and (inputs.requires_grad or self.weights.requires_grad or color_net.weights.requires_grad):
            w_s, ref_s, hook_s = self._train_weights()
            w_c, ref_c, hook_c = color_net._train_weights()
            dims_s = (self.input_dim, self.hidden_dim, self.num_layers, self.activation, self.output_activation)
            dims_c = (color_net.input_dim, color_net.hidden_dim, color_net.num_layers, color_net.activation, color_net.output_activation)
            return _FFMLPNgpPair.apply(inputs, w_s, w_c, dirs, dims_s, dims_c, (ref_s, ref_c), hook_s, hook_c,
                                       1 if level_major else 0, n_valid)
        return self._forward_ngp_pair_nograd(inputs, dirs, color_net, level_major, n_valid)

    @torch.no_grad()
    def _forward_ngp_pair_nograd(self, inputs, dirs, color_net, level_major=False, n_valid=None):
        B = inputs.shape[1] if level_major else inputs.shape[0]
        sigma = torch.empty(B, dtype=torch.float32, device=inputs.device)
        rgb = torch.empty(B, 3, dtype=torch.float32, device=inputs.device)
        _backend.ngp_pair_inference(inputs.contiguous(), self._inference_weights(), color_net._inference_weights(), B,
                                    self.hidden_dim, self.num_layers, color_net.num_layers, dirs.float().contiguous(), sigma, rgb,
                                    1 if level_major else 0, n_valid)
        return sigma, rgb

    def forward_rgb(self, inputs, n_valid=None):
        """the colour network with its head: fp32 [B, 3] = sigmoid(net(inputs)[:, :3]) from ONE launch (B % 128 == 0)"""
        if inputs.shape[0] % 128 != 0:
            raise RuntimeError("FFMLP.forward_rgb: needs a batch that is a multiple of 128 rows")
        return self._run(inputs, 0, n_valid, rgb_head=True)

    def forward_padded(self, inputs, level_major=False, n_valid=None):
        """forward() before the final column slice: [B, padded_output_dim] (columns >= output_dim are exact zeros'
        products: the padded weight rows).  Fused consumers (nerf/network_ff.py) read the 16-column rows directly.
        `level_major=True`: `inputs` is the grid encoder's [L, B, C] tensor (input_dim = L * C, C = 2, B % 128 == 0).
        `n_valid`: int32 GPU tensor, sample count of a padded batch (B % 128 == 0, training): rows past it (rounded up
        to 128) are skipped in both directions (seal3d_hip.h)."""
        if level_major:
            L, B, C = inputs.shape
            if C != 2 or L * C != self.input_dim or B % 128 != 0:
                raise RuntimeError("FFMLP level-major input: need [input_dim / 2, B, 2] with B % 128 == 0")
            return self._run(inputs, 1, n_valid)
        B, C = inputs.shape
        if n_valid is not None:
            if B % 128 != 0:
                raise RuntimeError("FFMLP: n_valid needs a batch that is a multiple of 128 rows")
            return self._run(inputs, 0, n_valid)
        # The reference always appends 128 - B % 128 zero rows (a full extra block when B is already aligned,
        # ffmlp.py:156-159) and slices them off again: results do not depend on it, and the copy is a full pass over the
        # activations, so rows are only added when the batch is ragged (the MFMA tiles are 32 points).
        pad = (-B) % 128
        if pad > 0:
            inputs = torch.cat([inputs, torch.zeros(pad, C, dtype=inputs.dtype, device=inputs.device)], dim=0)
        out = self._run(inputs, 0)
        if B != out.shape[0]:
            out = out[:B]
        return out

    def _run(self, inputs, input_layout, n_valid=None, rgb_head=False):
        w, ref, hook = self.weights, None, None
        if getattr(w, "_s3d_grad", None) is not None and getattr(w, "_s3d_half_version", None) == w._version:
            # a native optimizer maintains the fp16 copy of the weights and takes their gradient as an fp16 buffer
            if torch.is_grad_enabled() and w.requires_grad:
                ref = _ParamRef(w)
                # the fp16 copy carries no autograd history: a 0-dim CPU leaf (never cast, never read) keeps the node
                # in the graph when the inputs do not require a gradient either
                hook = self._autograd_hook
            w = w._s3d_half
        elif not torch.is_grad_enabled() and w.is_cuda:
            w = _cached_half(w)  # (inference without an optimizer: one cast per weight version, not one per call)
        out = ffmlp_forward(inputs, w, self.input_dim, self.padded_output_dim, self.hidden_dim,
                            self.num_layers, self.activation, self.output_activation, not self.training,
                            inputs.requires_grad, ref, hook, input_layout, n_valid, rgb_head)
        return out
