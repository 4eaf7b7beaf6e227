"""shencoder — drop-in for the reference's `shencoder` package (shencoder/sphere_harmonics.py)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import s3d_hip

_backend = s3d_hip.SHBackend


class _SHEncode(Function):
    """sphere_harmonics.py:14-54: fp32 always; the Jacobian is only produced when inputs need a gradient."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        n_out = degree ** 2
        outputs = torch.empty(B, n_out, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, D * n_out, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        _backend.sh_encode_forward(inputs, outputs, B, D, degree, dy_dx)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.meta = (B, D, degree)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, D, degree = ctx.meta
        grad_inputs = torch.zeros_like(inputs)
        _backend.sh_encode_backward(grad.contiguous(), inputs, B, D, degree, dy_dx, grad_inputs)
        return grad_inputs, None, None


sh_encode = _SHEncode.apply


class SHEncoder(nn.Module):
    """sphere_harmonics.py:61-87"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        lead = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        out = sh_encode(inputs, self.degree, inputs.requires_grad)
        return out.reshape(lead + [self.output_dim])
