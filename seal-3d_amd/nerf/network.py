"""NGP network used by Seal-3D (nerf/network.py of the reference): TWO hash encoders (density and colour),
degree-4 SH on the view direction, bias-free nn.Linear MLPs, `trunc_exp` density, sigmoid colour."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

import s3d_hip
from activation import trunc_exp
_ffb = s3d_hip.FFMLPBackend
from encoding import get_encoder
from ffmlp.ffmlp import _ParamRef, ffmlp_forward
from gridencoder.grid import grid_encode_pair

from .network_ff import _NgpRgb
from .renderer import NeRFRenderer

_head = s3d_hip.NgpHeadBackend


class _SealMid(Function):
    """sigma = trunc_exp(h[:, 0]);  colour-net input [B, 64] = [half(SH_4(d)) | h[:, 1:] | encoder_color(x) | 0] in one
    kernel per direction (nerf/network.py:106-126 of the reference: slice, exp, SH, cat with type promotion, cast);
    `enc_color` and its gradient stay in the grid kernels' level-major layout [16, B, 2]."""

    @staticmethod
    def forward(ctx, h, dirs, enc_color, n_valid=None):
        B = h.shape[0]
        sigma = torch.empty(B, dtype=torch.float32, device=h.device)
        cin = torch.empty(B, 64, dtype=torch.float16, device=h.device)
        _head.mid2_forward(h, dirs, enc_color, sigma, cin, n_valid)
        ctx.save_for_backward(h)
        ctx.n_valid = n_valid
        ctx.enc_grad = ctx.needs_input_grad[2]
        return sigma, cin

    @staticmethod
    def backward(ctx, g_sigma, g_cin):
        (h,) = ctx.saved_tensors
        B = h.shape[0]
        if g_cin is None:
            g_cin = torch.zeros(B, 64, dtype=torch.float16, device=h.device)
        g_h = torch.empty_like(h)
        # (rows past n_valid are never read downstream: the grid backward takes the same n_valid)
        g_enc = torch.empty(16, B, 2, dtype=torch.float16, device=h.device) if ctx.enc_grad else None
        _head.mid2_backward(g_cin.to(torch.float16).contiguous(), None if g_sigma is None else g_sigma.float().contiguous(), h, g_h,
                            g_enc, ctx.n_valid)
        return g_h, None, g_enc, None


class _SealPair(Function):
    """The whole network head of nerf/network.py:99-128 behind the two encoders — density MLP, trunc_exp, the colour-net input
    row [half(SH_4(d)) | h[:, 1:] | encoder_color(x) | 0], colour MLP, sigmoid — as ONE forward launch
    (s3d_ffmlp_ngp_pair_inference with enc_color: k_ffmlp_ngp_pair<true>; the row never leaves the chip in inference and is written
    once for the backward in training).  Backward = the calls the separate Functions make: colour MLP (colour head) -> mid2 backward
    -> density MLP, with ONE weight-gradient reduce launch for the two (s3d_ffmlp_wgrad_reduce_pair)."""

    @staticmethod
    def forward(ctx, e0, e1, w_sigma, w_color, dirs, refs, hook_s, hook_c, n_valid, keep):
        B = e0.shape[1]
        e0, e1 = e0.to(torch.half).contiguous(), e1.to(torch.half).contiguous()
        w_sigma, w_color = w_sigma.to(torch.half).contiguous(), w_color.to(torch.half).contiguous()
        dirs = dirs.float().contiguous()
        sigma = torch.empty(B, device=e0.device, dtype=torch.float32)
        rgb = torch.empty(B, 3, device=e0.device, dtype=torch.float32)
        cin = torch.empty(B, 64, device=e0.device, dtype=torch.half) if keep else None
        h0 = torch.empty(B, device=e0.device, dtype=torch.half) if keep else None
        _ffb.ngp_pair_inference(e0, w_sigma, w_color, B, 64, 2, 2, dirs, sigma, rgb, 1, n_valid, cin, h0, e1)
        if keep:
            ctx.save_for_backward(e0, w_sigma, w_color, h0, cin, rgb)
            ctx.refs, ctx.n_valid = refs, n_valid
            # (asked of autograd, not of the tensors: a cast made above under no-grad would hide an fp32 input's flag)
            ctx.need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
            ctx.set_materialize_grads(False)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        e0, w_sigma, w_color, h0, cin, rgb = ctx.saved_tensors
        B, dev = h0.shape[0], h0.device
        nv = {"n_valid": ctx.n_valid} if ctx.n_valid is not None else {}

        def target(ref, w):
            stash = getattr(ref.param, "_s3d_grad", None) if ref is not None else None
            acc, fi = False, None
            if stash is not None:
                acc = not getattr(ref.param, "_s3d_overwrite", False)
                fi = getattr(ref.param, "_s3d_found_inf", None)
            return stash, (stash.view(w.shape) if stash is not None else torch.empty_like(w)), acc, fi
        ref_s, ref_c = ctx.refs
        nb_c = (_ffb.backward_workspace_bytes(64, 16, 64, 2) + 255) // 256 * 256
        nb_s = _ffb.backward_workspace_bytes(32, 16, 64, 2)
        scratch = torch.empty(nb_c + nb_s, dtype=torch.uint8, device=dev)
        ws_c, ws_s = scratch[:nb_c], scratch[nb_c:]
        stash_c, gw_c, acc_c, fi_c = target(ref_c, w_color)
        g_cin = torch.empty_like(cin)
        if g_rgb is None:
            g_rgb = torch.zeros_like(rgb)
        ex_c = {"found_inf": fi_c} if fi_c is not None else {}
        _ffb.ffmlp_backward(None, cin, w_color, None, B, 64, 16, 64, 2, 0, 6, True, None, g_cin, gw_c,
                            grad_rgb=g_rgb.float().contiguous(), rgb_head=rgb, workspace=ws_c, defer_reduce=True, **nv, **ex_c)
        g_h = torch.empty(B, 16, dtype=torch.half, device=dev)
        g_e1 = torch.empty(16, B, 2, dtype=torch.half, device=dev) if ctx.need[1] else None
        _head.mid2_backward(g_cin, None if g_sigma is None else g_sigma.float().contiguous(), h0, g_h, g_e1, ctx.n_valid)
        stash_s, gw_s, acc_s, fi_s = target(ref_s, w_sigma)
        g_e0 = torch.empty_like(e0) if ctx.need[0] else torch.zeros(1, device=dev, dtype=e0.dtype)
        ex_s = {"found_inf": fi_s} if fi_s is not None else {}
        _ffb.ffmlp_backward(g_h, e0, w_sigma, None, B, 32, 16, 64, 2, 0, 6, ctx.need[0], None, g_e0, gw_s, input_layout=1,
                            workspace=ws_s, defer_reduce=True, **nv, **ex_s)
        _ffb.wgrad_reduce_pair((ws_c, B, 64, 64, 2, gw_c, acc_c, fi_c), (ws_s, B, 32, 64, 2, gw_s, acc_s, fi_s))
        for stash, ref in ((stash_s, ref_s), (stash_c, ref_c)):
            if stash is not None:
                ref.param._s3d_grad_touched = True
        return ((g_e0 if ctx.need[0] else None), g_e1, (None if stash_s is not None else gw_s),
                (None if stash_c is not None else gw_c)) + (None,) * 6


class PackedWeights:
    """The nn.Linear weights of one MLP inside the fused MLP kernels' flat fp16 layout ([out, in_padded] per layer, csrc/ffmlp.hip).

    The parameters stay what they are in the reference — names, shapes, fp32 values, optimizer state (checkpoint keys) — and this
    object holds their fp16 image `half` [numel]: member (parameter, offset, row stride) sits at rows `offset + r * stride`,
    constant blocks (an identity layer, zero padding) in between.  Two ways to keep it current:
      * adopted by nerf.optim.NativeAdam: the MLP backward writes the packed fp16 weight gradient into the pack's twin inside
        the optimizer's flat gradient buffer (`_s3d_grad`, overwritten each call), and the Adam launch reads it through a
        row-strided view per member and writes the updated fp16 weights into `half` through the same view — the step never
        assembles (cat / pad / cast) or splits anything;
      * otherwise (teacher, evaluation, frozen MLPs): `refreshed()` re-copies the members when one of them has changed (tensor
        version or optimizer epoch), i.e. once per weight update instead of once per call."""

    @property
    def _s3d_overwrite(self):
        """What the MLP backward asks before it writes the gradient twin.  The twin is never cleared (Adam reads it through the
        member views only, constant blocks would pile up under blind accumulation): the FIRST backward of a step replaces it,
        any further backward before the step — the two `density()` calls of `NeRFRenderer.run`, gradient accumulation over
        several batches — adds to it.  "Of a step": since the last zero_grad() / consuming step(), the same host-side flags
        the optimizer's own clearing decision uses (nerf/optim.py: zero_grad, clear_unconsumed)."""
        return not any(getattr(p, "_s3d_grad_touched", False) and not getattr(p, "_s3d_grad_consumed", False)
                       for p, _, _ in self.members)

    def __init__(self, members, numel, constants=()):
        self.members, self.numel, self.constants = list(members), int(numel), list(constants)
        self.half = None
        self.adopted = False
        self._s3d_grad = None
        self._key = None
        self.hook = torch.zeros((), requires_grad=True)  # keeps the autograd node alive when only the pack needs a gradient
        for p, off, stride in self.members:
            p._s3d_pack_spec = (self, off, stride)

    @staticmethod
    def usable_on(device):
        return device.type == "cuda"

    def _view(self, buf, p, off, stride):
        rows, cols = p.shape
        return buf[off:off + (rows - 1) * stride + cols].as_strided((rows, cols), (stride, 1))

    def _ensure(self, dev):
        if self.half is None or self.half.device != dev:
            self.half = torch.zeros(self.numel, dtype=torch.float16, device=dev)
            for off, block in self.constants:
                self.half[off:off + block.numel()].copy_(block.reshape(-1))
            self._key = None

    def adopt(self, grad_region, flat, flat_range):
        """nerf.optim.NativeAdam: `grad_region` = this pack's [numel] slice of the optimizer's flat fp16 gradient buffer"""
        self._ensure(grad_region.device)
        self._s3d_grad = grad_region
        for p, off, stride in self.members:
            rows, cols = p.shape
            p._s3d_grad = self._view(grad_region, p, off, stride)
            p._s3d_half = self._view(self.half, p, off, stride)
            p._s3d_half.copy_(p.detach())
            p._s3d_half_version = p._version
            p._s3d_grad_flat = flat
            p._s3d_flat_range = (flat_range[0] + off, flat_range[0] + off + (rows - 1) * stride + cols)
            p._s3d_grad_touched = False
            p._s3d_grad_consumed = False
        self.adopted = True

    # what the MLP backward reads / sets on the object it is handed as `param_ref.param` (ffmlp/ffmlp.py)
    @property
    def _s3d_found_inf(self):
        return getattr(self.members[0][0], "_s3d_found_inf", None)

    @property
    def _s3d_grad_touched(self):
        return any(getattr(p, "_s3d_grad_touched", False) for p, _, _ in self.members)

    @_s3d_grad_touched.setter
    def _s3d_grad_touched(self, v):
        for p, _, _ in self.members:
            if p.requires_grad or not v:
                p._s3d_grad_touched = v

    def trainable(self):
        return any(p.requires_grad for p, _, _ in self.members)

    def current(self):
        """the pack as maintained by the optimizer, or None when a member was written from outside since (or never adopted)"""
        if not self.adopted or self.half is None or self.half.device != self.members[0][0].device:
            return None
        if any(getattr(p, "_s3d_half_version", None) != p._version for p, _, _ in self.members):
            return None
        return self.half

    @torch.no_grad()
    def refreshed(self):
        """the pack re-copied from the parameters when one of them has changed (no-grad / frozen use)"""
        from gridencoder.grid import _weights_epoch_now
        dev = self.members[0][0].device
        self._ensure(dev)
        cur = self.current()
        if cur is not None:
            return cur
        key = tuple((p.data_ptr(), p._version) for p, _, _ in self.members) + (_weights_epoch_now(),)
        if key != self._key and not torch.cuda.is_current_stream_capturing():
            for p, off, stride in self.members:
                self._view(self.half, p, off, stride).copy_(p.detach())
            self._key = key
        elif key != self._key:  # (inside a capture a cached copy cannot be trusted and must not be cached)
            tmp = self.half.clone()
            for p, off, stride in self.members:
                self._view(tmp, p, off, stride).copy_(p.detach())
            return tmp
        return self.half


def _mlp(dims):
    return nn.ModuleList([nn.Linear(i, o, bias=False) for i, o in zip(dims[:-1], dims[1:])])


def _run_mlp(layers, h):
    for k, layer in enumerate(layers):
        h = layer(h)
        if k != len(layers) - 1:
            h = F.relu(h, inplace=True)
    return h


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, log2_hashmap_size=19, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound,
                                                log2_hashmap_size=log2_hashmap_size)
        self.sigma_net = _mlp([self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim])

        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        self.encoder_color, self.in_dim_color = get_encoder(encoding, desired_resolution=2048 * bound,
                                                            log2_hashmap_size=log2_hashmap_size)
        self.color_net = _mlp([self.in_dim_dir + geo_feat_dim + self.in_dim_color]
                              + [hidden_dim_color] * (num_layers_color - 1) + [3])
        if self.bg_radius > 0:
            raise NotImplementedError("background model (bg_radius > 0) is outside the BASELINE configs")
        self.register_buffer("_eye_hidden", torch.eye(hidden_dim), persistent=False)  # (not a checkpoint key)
        self._packs = None
        if (num_layers == 2 and hidden_dim == 64 and geo_feat_dim == 15 and self.in_dim == 32 and num_layers_color == 3
                and hidden_dim_color == 64 and self.in_dim_color == 32 and self.in_dim_dir == 16):
            s0, s1 = self.sigma_net[0].weight, self.sigma_net[1].weight
            c0, c1, c2 = (l.weight for l in self.color_net)
            # sigma pack [W0 64x32 | I 64x64 | W1 16x64]; colour pack [W0 64x63 in 64-wide rows | W1 64x64 | W2 3x64 in 16 rows]
            self._packs = (PackedWeights([(s0, 0, 32), (s1, 6144, 64)], 7168, [(2048, torch.eye(64))]),
                           PackedWeights([(c0, 0, 64), (c1, 4096, 64), (c2, 8192, 64)], 9216))

    # ---- MI355X path under `-O` (fp16 autocast): both MLPs run as fused MFMA kernels (csrc/ffmlp.hip) on weights PACKED
    # from the nn.Linear parameters — the parameters, their names and shapes (checkpoint keys) are the reference's.
    #   sigma net 32 -> 64 -> 16      = ffmlp [W0 | I_64 | W1]: relu(I relu(a)) == relu(a) exactly, so the inserted identity
    #                                    layer changes neither values nor gradients (its own gradient is discarded)
    #   colour net 63 -> 64 -> 64 -> 3 = ffmlp [W0 padded to 64 columns | W1 | W2 padded to 16 rows]
    # Same arithmetic as the reference's autocast path: fp16 operands, fp32 accumulation, fp16 activations; SH values
    # rounded to fp16 where the first Linear's input cast rounds them; sigmoid evaluated in fp32 and rounded to fp16.
    fused_pair = os.environ.get("S3D_FUSED_PAIR", "1") != "0"  # A-B runs: False = density MLP, head kernel and colour MLP as three launches
    fused_encoders = os.environ.get("S3D_FUSED_ENCODERS", "1") != "0"  # A-B runs: False = one forward launch per encoder
    fused_mlp = os.environ.get("S3D_FUSED_SEAL", "1") != "0"  # tests / A-B runs: False = nn.Linear op sequence

    def honours_row_limit(self, rows):
        return self._can_fuse_rows(self.density_bitfield.is_cuda, 2, rows)

    def _can_fuse(self, x):
        return self._can_fuse_rows(x.is_cuda, x.dim(), x.shape[0])

    def _can_fuse_rows(self, is_cuda, ndim, rows):
        return (self.fused_mlp and is_cuda and ndim == 2 and rows > 0 and rows % 128 == 0
                and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
                and self.num_layers == 2 and self.hidden_dim == 64 and self.geo_feat_dim == 15 and self.in_dim == 32
                and self.num_layers_color == 3 and self.hidden_dim_color == 64 and self.in_dim_color == 32
                and getattr(self.encoder_dir, "degree", None) == 4
                and getattr(self.encoder, "level_dim", 0) == 2 and getattr(self.encoder_color, "level_dim", 0) == 2)

    def _packed_weights(self):
        s0, s1 = self.sigma_net[0].weight, self.sigma_net[1].weight
        c0, c1, c2 = (l.weight for l in self.color_net)
        ws = torch.cat([s0.reshape(-1), self._eye_hidden.reshape(-1), s1.reshape(-1)])
        wc = torch.cat([F.pad(c0, (0, 1)).reshape(-1), c1.reshape(-1), F.pad(c2, (0, 0, 0, 13)).reshape(-1)])
        return ws, wc

    def _weights_for_kernels(self):
        """(sigma pack, colour pack, sigma ref, colour ref): the fp16 packed weights the MFMA kernels read, and — when the
        optimizer maintains them (PackedWeights.adopt) — the objects that receive their packed gradient"""
        packs = self._packs
        if packs is None:
            return self._packed_weights() + (None, None)
        grads = torch.is_grad_enabled()
        out, refs = [], []
        for pk in packs:
            need = grads and pk.trainable()
            cur = pk.current()
            if cur is not None:
                out.append(cur)
                refs.append(pk if need else None)
            elif not need:
                out.append(pk.refreshed())
                refs.append(None)
            else:
                return self._packed_weights() + (None, None)  # (torch optimizer: the differentiable cat / pad assembly)
        return out[0], out[1], refs[0], refs[1]

    def _forward_fused(self, x, d, want_rgb=True):
        nv = s3d_hip.active_row_limit(x.shape[0])  # (training: the march's sample count; inference: alive rays x n_step)
        live = None if (self.training or torch.is_grad_enabled()) else s3d_hip.active_live_rows(x.shape[0])
        ws, wc, rs, rc = self._weights_for_kernels()
        rs, hs = (None, None) if rs is None else (_ParamRef(rs), rs.hook)
        rc, hc = (None, None) if rc is None else (_ParamRef(rc), rc.hook)
        infer = not self.training
        # both encoders read the same points: one launch for the two tables when the colour is wanted too
        pair = grid_encode_pair(self.encoder, self.encoder_color, x, self.bound, nv, live) if (want_rgb and self.fused_encoders) else None
        e0 = pair[0] if pair is not None else self.encoder(x, bound=self.bound, level_major=True, n_valid=nv, live=live)
        if want_rgb and self.fused_pair and x.shape[0] % 128 == 0 and s3d_hip.FFMLPBackend.fused_backward_supported(64, 16, 64, 2, 0):
            # density MLP + head + colour MLP + sigmoid in one launch (the colour-net input row is built on chip)
            e1 = pair[1] if pair is not None else self.encoder_color(x, bound=self.bound, level_major=True, n_valid=nv, live=live)
            keep = torch.is_grad_enabled() and (e0.requires_grad or e1.requires_grad or ws.requires_grad or wc.requires_grad
                                                or rs is not None or rc is not None)
            return _SealPair.apply(e0, e1, ws, wc, d, (rs, rc), hs, hc, nv, keep)
        h = ffmlp_forward(e0, ws, 32, 16, 64, 2, 0, 6, infer, e0.requires_grad, rs, hs, 1, nv)
        if not want_rgb:
            return h
        e1 = pair[1] if pair is not None else self.encoder_color(x, bound=self.bound, level_major=True, n_valid=nv, live=live)
        sigma, cin = _SealMid.apply(h.contiguous(), d.float().contiguous(), e1.contiguous(), nv)
        if cin.shape[0] % 128 == 0 and (infer or s3d_hip.FFMLPBackend.fused_backward_supported(64, 16, 64, 2, 0)):
            # colour head inside the MLP kernels (seal3d_hip.h: rgb_head): fp32 sigmoid(out[:, :3]) straight from the last layer
            return sigma, ffmlp_forward(cin, wc, 64, 16, 64, 2, 0, 6, infer, cin.requires_grad, rc, hc, 0, nv, True)
        out = ffmlp_forward(cin, wc, 64, 16, 64, 2, 0, 6, infer, cin.requires_grad, rc, hc, 0, nv)
        return sigma, _NgpRgb.apply(out.contiguous(), nv)

    def _sigma(self, x):
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        return trunc_exp(h[..., 0]), h[..., 1:]

    def _rgb(self, x, d, geo_feat):
        h = torch.cat([self.encoder_dir(d), geo_feat, self.encoder_color(x, bound=self.bound)], dim=-1)
        return torch.sigmoid(_run_mlp(self.color_net, h))

    def forward(self, x, d):
        if self._can_fuse(x):
            return self._forward_fused(x, d)
        sigma, geo_feat = self._sigma(x)
        return sigma, self._rgb(x, d, geo_feat)

    def density(self, x):
        if self._can_fuse(x):
            h = self._forward_fused(x, None, want_rgb=False)
            return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}
        sigma, geo_feat = self._sigma(x)
        return {"sigma": sigma, "geo_feat": geo_feat}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is None:
            return self._rgb(x, d, geo_feat)
        rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
        if mask.any():
            rgbs[mask] = self._rgb(x[mask], d[mask], geo_feat[mask]).to(rgbs.dtype)
        return rgbs

    def get_params(self, lr):
        groups = [self.encoder, self.sigma_net, self.encoder_color, self.encoder_dir, self.color_net]
        return [{"params": g.parameters(), "lr": lr} for g in groups]
