"""NGP network used by Seal-3D (nerf/network.py of the reference): TWO hash encoders (density and colour),
degree-4 SH on the view direction, bias-free nn.Linear MLPs, `trunc_exp` density, sigmoid colour."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from activation import trunc_exp
from encoding import get_encoder

from .renderer import NeRFRenderer


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

    def _sigma(self, x):
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        return trunc_exp(h[..., 0]), h[..., 1:]

    def _rgb(self, x, d, geo_feat):
        h = torch.cat([self.encoder_dir(d), geo_feat, self.encoder_color(x, bound=self.bound)], dim=-1)
        return torch.sigmoid(_run_mlp(self.color_net, h))

    def forward(self, x, d):
        sigma, geo_feat = self._sigma(x)
        return sigma, self._rgb(x, d, geo_feat)

    def density(self, x):
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
