"""GPU parity: SH / frequency encoders vs golden vectors generated from the reference's own expressions
(tests/golden/sh_deg8.npz, oracle/gen_golden.py) and vs the CPU oracle.  Floating point: tolerances stated."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SH_RTOL, SH_ATOL = 2e-5, 1e-5  # fp32 recurrences vs fp32 expanded polynomials (values reach O(10) at degree 8)


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_vs_reference_expressions(hip, degree):
    g = np.load(os.path.join(GOLDEN, "sh_deg8.npz"))
    x = torch.from_numpy(g["inputs"]).cuda()
    B, n = x.shape[0], degree * degree
    out = torch.empty(B, n, device="cuda")
    jac = torch.empty(B, 3 * n, device="cuda")
    hip.SHBackend.sh_encode_forward(x, out, B, 3, degree, jac)
    torch.testing.assert_close(out.cpu(), torch.from_numpy(g["outputs"][:, :n]), rtol=SH_RTOL, atol=SH_ATOL)
    jac = jac.view(B, 3, n).cpu()
    for k, name in enumerate(("dx", "dy", "dz")):
        ref = torch.from_numpy(g[name][:, :n])
        torch.testing.assert_close(jac[:, k], ref, rtol=SH_RTOL, atol=SH_ATOL * max(1.0, float(ref.abs().max())))
    out2 = torch.empty(B, n, device="cuda")
    hip.SHBackend.sh_encode_forward(x, out2, B, 3, degree, None)   # no-Jacobian instantiation
    assert torch.equal(out, out2)


def test_sh_backward_vs_oracle(oracle, hip):
    g = torch.Generator().manual_seed(0)
    B, deg = 3001, 4
    x = torch.randn(B, 3, generator=g)
    x = x / x.norm(dim=-1, keepdim=True)
    grad = torch.randn(B, 16, generator=g)
    out_c, jac_c, gi_c = torch.empty(B, 16), torch.empty(B, 48), torch.zeros(B, 3)
    oracle.SHBackend.sh_encode_forward(x, out_c, B, 3, deg, jac_c)
    oracle.SHBackend.sh_encode_backward(grad, x, B, 3, deg, jac_c, gi_c)
    out_g, jac_g, gi_g = torch.empty(B, 16, device="cuda"), torch.empty(B, 48, device="cuda"), torch.zeros(B, 3, device="cuda")
    hip.SHBackend.sh_encode_forward(x.cuda(), out_g, B, 3, deg, jac_g)
    hip.SHBackend.sh_encode_backward(grad.cuda(), x.cuda(), B, 3, deg, jac_g, gi_g)
    torch.testing.assert_close(out_g.cpu(), out_c, rtol=SH_RTOL, atol=SH_ATOL)
    torch.testing.assert_close(gi_g.cpu(), gi_c, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("D,deg", [(3, 4), (27, 2), (3, 6), (1, 1)])
def test_freq_vs_oracle(oracle, hip, D, deg):
    g = torch.Generator().manual_seed(D)
    B, C = 2049, D + 2 * D * deg
    x = torch.rand(B, D, generator=g) * 2 - 1
    out_c = torch.empty(B, C)
    oracle.FreqBackend.freq_encode_forward(x, B, D, deg, C, out_c)
    out_g = torch.empty(B, C, device="cuda")
    hip.FreqBackend.freq_encode_forward(x.cuda(), B, D, deg, C, out_g)
    torch.testing.assert_close(out_g.cpu(), out_c, rtol=1e-5, atol=2e-6)
    # closed form: [x, sin(2^f x), cos(2^f x)]
    ref = [x]
    for f in range(deg):
        ref += [torch.sin(x.double() * 2 ** f).float(), torch.cos(x.double() * 2 ** f).float()]
    torch.testing.assert_close(out_g.cpu(), torch.cat(ref, -1), rtol=1e-5, atol=4e-6)
    grad = torch.randn(B, C, generator=g)
    gi_c, gi_g = torch.zeros(B, D), torch.zeros(B, D, device="cuda")
    oracle.FreqBackend.freq_encode_backward(grad, out_c, B, D, deg, C, gi_c)
    hip.FreqBackend.freq_encode_backward(grad.cuda(), out_g, B, D, deg, C, gi_g)
    torch.testing.assert_close(gi_g.cpu(), gi_c, rtol=1e-4, atol=1e-5)
