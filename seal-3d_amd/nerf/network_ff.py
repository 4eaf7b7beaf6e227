"""NGP network with fully fused MLPs (nerf/network_ff.py of the reference, `main_nerf.py --ff`):
hashgrid(32) -> FFMLP 32-64-64?-16 (density + 15 geo features) ; [SH16 | geo15 | 0] (32) -> FFMLP 32-64-64-3."""
import torch

from activation import trunc_exp
from encoding import get_encoder
from ffmlp import FFMLP

from .renderer import NeRFRenderer


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
        h = self.sigma_net(self.encoder(x, bound=self.bound))
        return trunc_exp(h[..., 0]), h[..., 1:]

    def _rgb(self, d, geo_feat):
        d = self.encoder_dir(d)
        pad = torch.zeros_like(geo_feat[..., :1])
        return torch.sigmoid(self.color_net(torch.cat([d, geo_feat, pad], dim=-1)))

    def forward(self, x, d):
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
