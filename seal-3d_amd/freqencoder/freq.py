"""freqencoder — drop-in for the reference's `freqencoder` package (freqencoder/freq.py)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import s3d_hip

_backend = s3d_hip.FreqBackend


def _on_device(t):
    if getattr(_backend, "device_type", "cuda") == "cuda" and not t.is_cuda:
        return t.cuda()
    return t


class _FreqEncode(Function):
    """freq.py:15-49: [x, sin(2^f x), cos(2^f x)]_{f<degree}; backward from the saved outputs."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        inputs = _on_device(inputs).contiguous()
        B, D = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        _backend.freq_encode_forward(inputs, B, D, degree, output_dim, outputs)
        ctx.save_for_backward(inputs, outputs)
        ctx.meta = (B, D, degree, output_dim)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, outputs = ctx.saved_tensors
        B, D, degree, output_dim = ctx.meta
        grad_inputs = torch.zeros_like(inputs)
        _backend.freq_encode_backward(grad.contiguous(), outputs, B, D, degree, output_dim, grad_inputs)
        return grad_inputs, None, None


freq_encode = _FreqEncode.apply


class FreqEncoder(nn.Module):
    """freq.py:55-77"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        lead = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        out = freq_encode(inputs, self.degree, self.output_dim)
        return out.reshape(lead + [self.output_dim])
