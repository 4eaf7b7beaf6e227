"""CPU: oracle/vm_features.py (numpy restatement of tensoRF/network.py:112-153 + torch's grid sampler) pinned against the
reference's own op sequence — twelve F.grid_sample calls on CPU — values and autograd gradients; and the module's CPU path."""
import numpy as np
import torch

from oracle import vm_features as vo
from tensoRF import network as trf


def _setup():
    torch.manual_seed(5)
    net = trf.NeRFNetwork(resolution=[12, 20, 16], sigma_rank=[3, 2, 4], color_rank=[5, 6, 2], bound=1, cuda_ray=True)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(3000, 3, generator=g) * 2.6 - 1.3
    x[:8] = torch.tensor([[-1.0, 1.0, 0.0]])
    x[8:16] = 1.0
    return net, x


def _np(params):
    return [p.detach().numpy()[0] if p.shape[-1] != 1 else p.detach().numpy()[0, :, :, 0] for p in params]


def test_oracle_matches_grid_sample_sequence():
    net, x = _setup()
    xs = x.numpy()
    s_ref = net._sigma_feat_torch(x, net.sigma_mat, net.sigma_vec)
    c_ref = net._color_prod_torch(x, net.color_mat, net.color_vec)
    s = vo.sigma_feat(xs, _np(net.sigma_mat), _np(net.sigma_vec))
    c = vo.color_products(xs, _np(net.color_mat), _np(net.color_vec))
    np.testing.assert_allclose(s, s_ref.detach().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(c, c_ref.detach().numpy(), rtol=1e-5, atol=1e-7)
    inside = (x.abs() <= 1).all(1).numpy()
    assert 0.2 < inside.mean() < 0.8 and np.abs(s[~inside]).max() > 0
    # gradients: autograd through grid_sample vs the restated scatter-add
    w = torch.linspace(-1, 1, x.shape[0])
    s_ref.mul(w).sum().backward()
    gp, gl = vo.factor_grads(xs, _np(net.sigma_mat), _np(net.sigma_vec), np.tile(w.numpy()[None], (9, 1)))
    for p, g in zip(list(net.sigma_mat) + list(net.sigma_vec), gp + gl):
        np.testing.assert_allclose(g.reshape(p.grad.shape), p.grad.numpy(), rtol=2e-4, atol=1e-6)
    gc = torch.randn(c_ref.shape, generator=torch.Generator().manual_seed(7))
    c_ref.mul(gc).sum().backward()
    gp, gl = vo.factor_grads(xs, _np(net.color_mat), _np(net.color_vec), gc.numpy())
    for p, g in zip(list(net.color_mat) + list(net.color_vec), gp + gl):
        np.testing.assert_allclose(g.reshape(p.grad.shape), p.grad.numpy(), rtol=2e-4, atol=1e-6)


def test_module_cpu_path_is_the_reference_sequence(oracle_wrappers):
    net, x = _setup()
    assert not net._use_native(x)  # CPU tensors: the grid_sample sequence, exactly as in the reference
    sigma, rgb = net(x, torch.nn.functional.normalize(torch.randn(x.shape[0], 3), dim=-1))
    assert sigma.shape == (3000,) and rgb.shape == (3000, 3) and torch.isfinite(rgb).all()


def test_oracle_edge_resolutions_and_border_points():
    """resolution 2 along an axis, points exactly on -1 / +1 / 0 and far outside: the restatement still equals grid_sample"""
    import itertools
    torch.manual_seed(11)
    for res in ([2, 5, 3], [7, 2, 2], [4, 4, 9]):
        net = trf.NeRFNetwork(resolution=res, sigma_rank=[2, 1, 3], color_rank=[1, 2, 2], bound=1, cuda_ray=True)
        corners = torch.tensor(list(itertools.product([-1.0, 0.0, 1.0, -1.0001, 1.0001, 3.0], repeat=3)))
        x = torch.cat([corners, torch.rand(200, 3) * 2 - 1])
        s_ref = net._sigma_feat_torch(x, net.sigma_mat, net.sigma_vec).detach().numpy()
        c_ref = net._color_prod_torch(x, net.color_mat, net.color_vec).detach().numpy()
        np.testing.assert_allclose(vo.sigma_feat(x.numpy(), _np(net.sigma_mat), _np(net.sigma_vec)), s_ref, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(vo.color_products(x.numpy(), _np(net.color_mat), _np(net.color_vec)), c_ref, rtol=1e-5, atol=1e-7)


def test_tall_linear_chunked_weight_gradient_equals_autograd_on_cpu():
    """tensoRF/network.py:_TallLinear (weight gradient of a bias-free Linear as a batched GEMM over 1,024-row chunks + a remainder
    GEMM, summed in fp32) is the same function as F.linear under autograd: values, data gradient and weight gradient on CPU
    tensors in fp32 (the GPU test compares it with the fp16 autocast nn.Linear), ragged row counts included; and `_linear` only
    takes that route on the GPU under autocast."""
    for N in (5 * 1024, 5 * 1024 + 333, 700):
        torch.manual_seed(N)
        layer = torch.nn.Linear(24, 7, bias=False)
        x0, go = torch.randn(N, 24), torch.randn(N, 7)
        res = []
        for tall in (True, False):
            layer.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = trf._TallLinear.apply(x, layer.weight) if tall else layer(x)
            y.backward(go)
            res.append((y.detach(), x.grad.clone(), layer.weight.grad.clone()))
        torch.testing.assert_close(res[0][0], res[1][0], rtol=0, atol=0)
        torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-5, atol=1e-4)
    y = trf._linear(layer, x0)  # CPU tensors: the module itself
    assert y.grad_fn is None or "TallLinear" not in type(y.grad_fn).__name__
