"""Density activation of the NGP networks.

`trunc_exp(h)` is exp(h) whose derivative is evaluated at a clamped argument, so one runaway pre-activation cannot put
an inf into the hash-table gradient while the forward value stays the plain exponential (semantics of the reference's
activation.py:5-17).  Under autocast it is computed in fp32: exp(11.1) already overflows fp16.

The fused NGP head (csrc/ngp_head.hip, k_ngp_mid_forward/backward) evaluates the same two expressions in its kernels;
this module is the stand-alone op for every other call site (density queries, the nn.Linear network, TensoRF).
"""
import torch

_GRAD_CLAMP = 15.0  # |argument| limit of the derivative: exp(15) ~ 3.3e6


class _TruncatedExp(torch.autograd.Function):
    # new-style Function: forward is context-free, the context is filled in separately
    @staticmethod
    def forward(pre_activation, compute_dtype):
        return torch.exp(pre_activation.to(compute_dtype))

    @staticmethod
    def setup_context(ctx, inputs, output):
        pre_activation, compute_dtype = inputs
        ctx.save_for_backward(pre_activation)
        ctx.compute_dtype = compute_dtype

    @staticmethod
    def backward(ctx, grad_output):
        (pre_activation,) = ctx.saved_tensors
        slope = torch.exp(pre_activation.to(ctx.compute_dtype).clamp(min=-_GRAD_CLAMP, max=_GRAD_CLAMP))
        return (grad_output * slope).to(pre_activation.dtype), None


def trunc_exp(pre_activation):
    """exp with the clamped derivative.  Under autocast the op runs in fp32 (the reference registers it with
    `cast_inputs=torch.float`), otherwise in the dtype it is given."""
    kind = pre_activation.device.type
    autocast_on = torch.is_autocast_enabled(kind) if kind in ("cuda", "cpu") else False
    compute_dtype = torch.float32 if autocast_on else pre_activation.dtype
    with torch.autocast(kind, enabled=False):
        return _TruncatedExp.apply(pre_activation, compute_dtype)
