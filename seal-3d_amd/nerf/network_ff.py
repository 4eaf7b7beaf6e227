"""NGP network with fully fused MLPs (nerf/network_ff.py of the reference, `main_nerf.py --ff`):
hashgrid(32) -> FFMLP 32-64-64?-16 (density + 15 geo features) ; [SH16 | geo15 | 0] (32) -> FFMLP 32-64-64-3."""
import os

import torch
from torch.autograd import Function

import s3d_hip
from activation import trunc_exp
from encoding import get_encoder
from ffmlp import FFMLP

from .renderer import NeRFRenderer


_head = s3d_hip.NgpHeadBackend


class _NgpMid(Function):
    """sigma = trunc_exp(h[:, 0]);  colour-net input = [half(SH_4(d)) | h[:, 1:] | 0]  in one kernel per direction
    (network_ff.py:55-96 of the reference does this with slice, exp, SH, zeros, cat and cast nodes)."""

    @staticmethod
    def forward(ctx, h, dirs, n_valid=None):
        B = h.shape[0]
        sigma = torch.empty(B, dtype=torch.float32, device=h.device)
        cin = torch.empty(B, 32, dtype=torch.float16, device=h.device)
        _head.mid_forward(h, dirs, sigma, cin, n_valid)
        ctx.save_for_backward(h)
        ctx.n_valid = n_valid
        return sigma, cin

    @staticmethod
    def backward(ctx, g_sigma, g_cin):
        (h,) = ctx.saved_tensors
        if g_cin is None:
            g_cin = torch.zeros(h.shape[0], 32, dtype=torch.float16, device=h.device)
        g_h = torch.empty_like(h)
        _head.mid_backward(g_cin.to(torch.float16).contiguous(), None if g_sigma is None else g_sigma.float().contiguous(), h, g_h,
                           ctx.n_valid)
        return g_h, None, None


class _NgpRgb(Function):
    """rgb = sigmoid(colour_net_output[:, :3]) as fp32, with fp16 rounding where torch.sigmoid on the fp16 tensor rounds"""

    @staticmethod
    def forward(ctx, out, n_valid=None):
        rgb = torch.empty(out.shape[0], 3, dtype=torch.float32, device=out.device)
        _head.rgb_forward(out, rgb, n_valid)
        ctx.save_for_backward(rgb)
        ctx.n_valid = n_valid
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        (rgb,) = ctx.saved_tensors
        g_out = torch.empty(rgb.shape[0], 16, dtype=torch.float16, device=rgb.device)
        _head.rgb_backward(g_rgb.float().contiguous(), rgb, g_out, ctx.n_valid)
        return g_out, None


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + geo_feat_dim, hidden_dim=hidden_dim,
                               num_layers=num_layers)
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder_dir, self.in_dim_color = get_encoder(encoding_dir)
        self.in_dim_color += geo_feat_dim + 1  # pad 31 -> 32 (network_ff.py:44)
        self.color_net = FFMLP(input_dim=self.in_dim_color, output_dim=3, hidden_dim=hidden_dim_color,
                               num_layers=num_layers_color)

    def _sigma(self, x):
        if self._can_fuse(x) and x.dim() == 2 and x.shape[0] > 0 and x.shape[0] % 128 == 0:
            # level-major hand-over (no [L,B,C] -> [B,L*C] permute copy); same values
            h = self.sigma_net.forward_padded(self.encoder(x, bound=self.bound, level_major=True), level_major=True)
        else:
            h = self.sigma_net(self.encoder(x, bound=self.bound))
        return trunc_exp(h[..., 0]), h[..., 1:]

    def _rgb(self, d, geo_feat):
        d = self.encoder_dir(d)
        pad = torch.zeros_like(geo_feat[..., :1])
        return torch.sigmoid(self.color_net(torch.cat([d, geo_feat, pad], dim=-1)))

    fused_head = os.environ.get("S3D_FUSED_HEAD", "1") != "0"  # tests / A-B runs: False = the reference op sequence
    fused_pair = os.environ.get("S3D_FUSED_PAIR", "1") != "0"  # A-B runs: False = one launch per network in the inference loop
    fused_mid = os.environ.get("S3D_FUSED_MID", "1") != "0"    # A-B runs: False = separate mid kernels between the two MLPs

    def honours_row_limit(self, rows):
        return self._can_fuse_on(self.density_bitfield.is_cuda) and rows > 0 and rows % 128 == 0

    def _can_fuse(self, x):
        return self._can_fuse_on(x.is_cuda)

    def _can_fuse_on(self, is_cuda):
        return (self.fused_head and is_cuda and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.float16
                and getattr(self.encoder_dir, "degree", None) == 4 and self.geo_feat_dim == 15
                and self.sigma_net.padded_output_dim == 16 and self.color_net.input_dim == 32
                and self.color_net.padded_output_dim == 16
                # widths the fused MFMA kernels do not cover (hidden 16 / 256, wide inputs: the layer-by-layer MFMA kernels of csrc/ffmlp_generic.hip) take the reference's
                # op sequence: the level-major / n_valid / head routes exist in the fused kernels only
                and self._fused_mlps())

    def _fused_mlps(self):
        ok = getattr(self, "_fused_mlps_ok", None)
        if ok is None:
            ok = self._fused_mlps_ok = bool(self.sigma_net.fused_supported() and self.color_net.fused_supported())
        return ok

    def forward(self, x, d):
        if self._can_fuse(x):
            if x.dim() == 2 and x.shape[0] % 128 == 0:  # level-major hand-over: no permute copy in either direction
                # every op from here to sigma / rgb is a native kernel, so a padded training batch announced by the renderer
                # (s3d_hip.row_limit: the march's device-side sample count) is honoured end to end: the absent tail of
                # the buffers is neither computed nor back-propagated
                nv = s3d_hip.active_row_limit(x.shape[0])  # (training: the march's sample count; inference: alive rays x n_step)
                # inference loop: unused sample slots (deltas == 0, announced by the renderer) skip the table gathers
                live = None if (self.training or torch.is_grad_enabled()) else s3d_hip.active_live_rows(x.shape[0])
                enc = self.encoder(x, bound=self.bound, level_major=True, n_valid=nv, live=live)
                if (self.fused_pair and self.fused_mid and self.sigma_net.rgb_head_supported()
                        and self.color_net.rgb_head_supported() and self.sigma_net.pair_supported(self.color_net)):
                    # both networks and both heads in one forward launch (inference: the colour-net input never leaves the chip;
                    # training: it is written once for the backward instead of written and read back)
                    return self.sigma_net.forward_ngp_pair(enc, d, self.color_net, level_major=True, n_valid=nv)
                if self.fused_mid and self.sigma_net.rgb_head_supported():  # (same shape condition: the fused backward kernel)
                    # trunc_exp / SH / concat folded into the density network's last layer (and its backward's first load)
                    sigma, cin = self.sigma_net.forward_ngp_mid(enc, d, level_major=True, n_valid=nv)
                else:
                    h = self.sigma_net.forward_padded(enc, level_major=True, n_valid=nv)
                    sigma, cin = _NgpMid.apply(h.contiguous(), d.float().contiguous(), nv)
                if self.color_net.rgb_head_supported():  # sigmoid + fp32 hand-over inside the last layer's store
                    return sigma, self.color_net.forward_rgb(cin, n_valid=nv)
                return sigma, _NgpRgb.apply(self.color_net.forward_padded(cin, n_valid=nv).contiguous(), nv)
            h = self.sigma_net.forward_padded(self.encoder(x, bound=self.bound))
            sigma, cin = _NgpMid.apply(h.contiguous(), d.float().contiguous())
            if cin.shape[0] % 128 == 0 and self.color_net.rgb_head_supported():
                return sigma, self.color_net.forward_rgb(cin)
            return sigma, _NgpRgb.apply(self.color_net.forward_padded(cin).contiguous())
        sigma, geo_feat = self._sigma(x)
        return sigma, self._rgb(d, geo_feat)

    def density(self, x):
        sigma, geo_feat = self._sigma(x)
        return {"sigma": sigma, "geo_feat": geo_feat}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is None:
            return self._rgb(d, geo_feat)
        rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
        if mask.any():
            rgbs[mask] = self._rgb(d[mask], geo_feat[mask]).to(rgbs.dtype)
        return rgbs

    def get_params(self, lr):
        groups = [self.encoder, self.sigma_net, self.encoder_dir, self.color_net]
        return [{"params": g.parameters(), "lr": lr} for g in groups]
