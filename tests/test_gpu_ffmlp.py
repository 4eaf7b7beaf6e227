"""GPU parity: MFMA fused MLP vs the CPU oracle (dense math, fp16 storage / fp32 accumulate) and vs the
bias-free torch MLP twin the reference's own test uses (testing/test_ffmlp.py:11-43).
fp16 tolerance: one half ulp per layer boundary can flip => rtol 2e-2 / atol 2e-2 on O(1) activations."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

NETS = [  # in, W, n_layers, out(real), activation
    (32, 64, 2, 16, 0),   # sigma net of nerf/network_ff.py
    (32, 64, 3, 3, 0),    # colour net
    (16, 64, 2, 16, 0),   # testing/test_ffmlp.py shape
    (64, 64, 2, 8, 0),    # in == W
    (32, 32, 4, 16, 0),   # W = 32, deeper
    (48, 64, 2, 16, 3),   # sigmoid, in = 48
    # W = 128: the MFMA kernels of the stored-activation path (forward / dgrad / wgrad with 64 KiB partial planes, round 5)
    (32, 128, 3, 16, 0),
    (32, 128, 2, 16, 3),  # sigmoid
    (128, 128, 2, 16, 0),  # in == W = 128
    (64, 128, 4, 5, 0),   # three hidden matrices: 136 KiB of weight fragments per workgroup
    (160, 128, 3, 3, 0),  # input wider than the hidden width (TensoRF's colour MLP, 150 padded to 160): first weight gradient in two column blocks
    (144, 128, 2, 16, 0),
    # the other widths ffmlp.cu:40-44 dispatches, and a 64-wide network with an input wider than 64: layer-by-layer path
    # (csrc/ffmlp_generic.hip)
    (32, 16, 2, 16, 0),
    (64, 256, 2, 3, 0),
    (128, 64, 2, 16, 0),
]


def _weights(in_dim, W, n, seed=42):
    torch.manual_seed(seed)
    num = W * (in_dim + W * (n - 1) + 16)
    std = math.sqrt(3 / W)
    return torch.empty(num).uniform_(-std, std)


def _split(w, in_dim, W, n):
    mats, off = [], 0
    for k, o in [(in_dim, W)] + [(W, W)] * (n - 1) + [(W, 16)]:
        mats.append(w[off:off + o * k].view(o, k))
        off += o * k
    return mats


@pytest.mark.parametrize("in_dim,W,n,out,act", NETS)
@pytest.mark.parametrize("B", [128, 128 * 37, 128 * 2100])  # the last: training size, several tiles per wave / workgroup
def test_ffmlp_forward_backward(oracle, hip, in_dim, W, n, out, act, B):
    if B > 128 * 37 and (in_dim, W, n) not in ((32, 64, 2), (32, 64, 3)):
        pytest.skip("training-size batch only for the two networks of nerf/network_ff.py")
    g = torch.Generator().manual_seed(B + in_dim)
    w = _weights(in_dim, W, n)
    mats = _split(w, in_dim, W, n)
    for m in mats[-1:]:
        m[out:] = 0  # padded output rows
    w16 = w.half()
    x = (torch.randn(B, in_dim, generator=g) * 0.5).half()
    # ---- oracle
    fb_c = torch.empty(n, B, W, dtype=torch.half)
    out_c = torch.empty(B, 16, dtype=torch.half)
    oracle.FFMLPBackend.ffmlp_forward(x, w16, B, in_dim, 16, W, n, act, 6, fb_c, out_c)
    # ---- hip
    xg, wg = x.cuda(), w16.cuda()
    fb_g = torch.empty(n, B, W, dtype=torch.half, device="cuda")
    out_g = torch.empty(B, 16, dtype=torch.half, device="cuda")
    hip.FFMLPBackend.ffmlp_forward(xg, wg, B, in_dim, 16, W, n, act, 6, fb_g, out_g)
    out_i = torch.empty(B, 16, dtype=torch.half, device="cuda")
    hip.FFMLPBackend.ffmlp_inference(xg, wg, B, in_dim, 16, W, n, act, 6, torch.empty(2, B, W, dtype=torch.half, device="cuda"), out_i)
    torch.cuda.synchronize()
    assert torch.equal(out_g, out_i), "training and inference kernels must agree exactly"
    torch.testing.assert_close(out_g.cpu().float(), out_c.float(), rtol=2e-2, atol=2e-2)
    # ---- torch twin (fp32 math on the fp16-rounded weights)
    h = x.float()
    for i, m in enumerate(_split(w16.float(), in_dim, W, n)):
        h = h @ m.t()
        if i != n:
            h = torch.relu(h) if act == 0 else torch.sigmoid(h)
    torch.testing.assert_close(out_g.cpu().float(), h, rtol=3e-2, atol=3e-2)

    # ---- backward
    grad = (torch.randn(B, 16, generator=g) * 0.1).half()
    grad[:, out:] = 0
    bb_c = torch.zeros(n, B, W, dtype=torch.half)
    gi_c = torch.zeros(B, in_dim, dtype=torch.half)
    gw_c = torch.zeros_like(w16)
    gw32 = oracle.FFMLPBackend.ffmlp_backward(grad, x, w16, fb_c, B, in_dim, 16, W, n, act, 6, True, bb_c, gi_c, gw_c)
    bb_g = torch.zeros(n, B, W, dtype=torch.half, device="cuda")
    gi_g = torch.zeros(B, in_dim, dtype=torch.half, device="cuda")
    gw_g = torch.zeros_like(wg)
    hip.FFMLPBackend.ffmlp_backward(grad.cuda(), xg, wg, fb_g, B, in_dim, 16, W, n, act, 6, True, bb_g, gi_g, gw_g)
    torch.cuda.synchronize()
    # ReLU gates of near-zero pre-activations may flip between accumulation orders: L2 comparison
    assert (gi_g.cpu().float() - gi_c.float()).norm() / gi_c.float().norm().clamp(min=1e-6) < 2e-2
    scale = gw32.abs().max().clamp(min=1e-3)
    err = (gw_g.cpu().float() - gw32).abs().max() / scale
    assert err < 2e-2, f"weight-gradient relative error {err}"

    # ---- fused backward (no forward_buffer / backward_buffer: activations re-computed in the kernel).  Same MFMA
    # operations on the same values for the data gradient => bit-identical grad_inputs; the weight gradient sums the
    # same products in a different (fixed) order => fp32 rounding only, and run-to-run reproducible.
    if hip.FFMLPBackend.fused_backward_supported(in_dim, 16, W, n, act):
        runs = []
        for _ in range(2):
            gi_f = torch.zeros(B, in_dim, dtype=torch.half, device="cuda")
            gw_f = torch.zeros_like(wg)
            hip.FFMLPBackend.ffmlp_backward(grad.cuda(), xg, wg, None, B, in_dim, 16, W, n, act, 6, True, None, gi_f, gw_f)
            runs.append((gi_f.cpu(), gw_f.cpu()))
        assert torch.equal(runs[0][0].view(torch.int16), gi_g.cpu().view(torch.int16)), "fused dgrad differs from the two-kernel path"
        assert torch.equal(runs[0][1].view(torch.int16), runs[1][1].view(torch.int16)), "fused wgrad must be reproducible"
        errf = (runs[0][1].float() - gw32).abs().max() / scale
        assert errf < 2e-2, f"fused weight-gradient relative error {errf}"
        assert (runs[0][1].float() - gw_g.cpu().float()).abs().max() / scale < 4e-3
        gw_n = torch.zeros_like(wg)  # weights only (no grad_inputs): the first-layer transposed fragments are skipped
        hip.FFMLPBackend.ffmlp_backward(grad.cuda(), xg, wg, None, B, in_dim, 16, W, n, act, 6, False, None, None, gw_n)
        assert torch.equal(gw_n.cpu().view(torch.int16), runs[0][1].view(torch.int16))
    else:
        assert W == 64 and n - 1 > 2 or act == 2 or W not in (32, 64) or in_dim > 64


def test_ffmlp_rejects_bad_shapes(hip):
    x = torch.zeros(100, 32, dtype=torch.half, device="cuda")
    w = torch.zeros(64 * (32 + 64 + 16), dtype=torch.half, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 128"):
        hip.FFMLPBackend.ffmlp_forward(x, w, 100, 32, 16, 64, 2, 0, 6, None, torch.empty(100, 16, dtype=torch.half, device="cuda"))
    x = torch.zeros(128, 32, dtype=torch.half, device="cuda")
    with pytest.raises(RuntimeError, match="only support hidden_dim"):  # (the reference's message, ffmlp.cu:44)
        hip.FFMLPBackend.ffmlp_forward(x, w, 128, 32, 16, 48, 2, 0, 6, None, torch.empty(128, 16, dtype=torch.half, device="cuda"))
    with pytest.raises(RuntimeError, match="needs forward_buffer"):  # hidden 256: the layer-by-layer path needs its activation buffers
        hip.FFMLPBackend.ffmlp_forward(x, torch.zeros(256 * (32 + 256 + 16), dtype=torch.half, device="cuda"), 128, 32, 16, 256, 2, 0, 6, None,
                                       torch.empty(128, 16, dtype=torch.half, device="cuda"))
    # hidden 128 runs on the MFMA kernels: a forward without forward_buffer is an inference call, as for 32 / 64
    hip.FFMLPBackend.ffmlp_forward(x, torch.zeros(128 * (32 + 128 + 16), dtype=torch.half, device="cuda"), 128, 32, 16, 128, 2, 0, 6, None,
                                   torch.empty(128, 16, dtype=torch.half, device="cuda"))
    assert not hip.FFMLPBackend.fused_backward_supported(32, 16, 128, 2, 0)   # ... and trains through forward_buffer / backward_buffer


def test_ffmlp_level_major_input_layout(hip):
    """input_layout=1 reads the grid encoder's [L, B, 2] tensor in place and writes grad_inputs in the same layout: results
    must be bit-identical to the row-major call on the permuted data (same fragments, same MFMA sequence)."""
    in_dim, W, n, B = 32, 64, 2, 128 * 21
    g = torch.Generator().manual_seed(5)
    w = _weights(in_dim, W, n).half().cuda()
    x = (torch.randn(B, in_dim, generator=g) * 0.5).half().cuda()
    xl = x.view(B, in_dim // 2, 2).permute(1, 0, 2).contiguous()                 # [L, B, 2]
    grad = (torch.randn(B, 16, generator=g) * 0.1).half().cuda()
    F = hip.FFMLPBackend
    outs = []
    for layout, xin in ((0, x), (1, xl)):
        out = torch.empty(B, 16, dtype=torch.half, device="cuda")
        F.ffmlp_forward(xin, w, B, in_dim, 16, W, n, 0, 6, None, out, input_layout=layout)
        gi = torch.empty_like(xin)
        gw = torch.zeros_like(w)
        F.ffmlp_backward(grad, xin, w, None, B, in_dim, 16, W, n, 0, 6, True, None, gi, gw, input_layout=layout)
        outs.append((out, gi if layout == 0 else gi.permute(1, 0, 2).reshape(B, in_dim), gw))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int16), b.contiguous().view(torch.int16))


@pytest.mark.parametrize("B", [128, 128 * 700])
def test_colour_head_in_the_last_layer_is_the_two_kernel_sequence_bit_for_bit(hip, B):
    """FFMLP.forward_rgb (seal3d_hip.h: rgb_head) == forward_padded + the separate sigmoid head kernels of nerf/network_ff.py:
    fp32 [B, 3] output, weight gradient and input gradient, training and inference, with and without a device row count."""
    from ffmlp import FFMLP
    from nerf.network_ff import _NgpRgb
    torch.manual_seed(5)
    net = FFMLP(32, 3, 64, 3).cuda()
    assert net.rgb_head_supported()
    x0 = torch.randn(B, 32, device="cuda").half()
    g = torch.randn(B, 3, device="cuda")
    for nv in (None, torch.tensor([max(B - 300, 77)], dtype=torch.int32, device="cuda")):
        rows = B if nv is None else min(B, (int(nv) + 127) // 128 * 128)
        res = []
        for fused in (False, True):
            net.train()
            net.zero_grad()
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.float16):
                rgb = net.forward_rgb(x, n_valid=nv) if fused else _NgpRgb.apply(net.forward_padded(x, n_valid=nv).contiguous(), nv)
            assert rgb.dtype == torch.float32 and rgb.shape == (B, 3)
            (rgb[:rows] * g[:rows]).sum().backward()
            res.append((rgb.detach()[:rows].clone(), net.weights.grad.clone(), x.grad[:rows].clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        assert float(res[0][1].abs().max()) > 0
        net.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            a = net.forward_rgb(x0, n_valid=nv)[:rows]
            b = _NgpRgb.apply(net.forward_padded(x0, n_valid=nv).contiguous(), nv)[:rows]
        assert torch.equal(a, b) and torch.equal(a, res[0][0])


@pytest.mark.parametrize("B", [128, 128 * 700])
def test_density_head_in_the_mlp_kernels_is_the_mid_kernel_sequence_bit_for_bit(hip, B):
    """FFMLP.forward_ngp_mid (seal3d_hip.h: mid_*) == forward_padded + k_ngp_mid_forward / _backward of nerf/network_ff.py:
    sigma, colour-net input, weight gradient and input gradient; training and inference; with a device row count."""
    from ffmlp import FFMLP
    from nerf.network_ff import _NgpMid
    torch.manual_seed(7)
    net = FFMLP(32, 16, 64, 2).cuda()
    x0 = torch.randn(16, B, 2, device="cuda").half()  # level-major encoder output
    d = torch.nn.functional.normalize(torch.randn(B, 3, device="cuda"), dim=-1)
    gs, gc = torch.randn(B, device="cuda"), (torch.randn(B, 32, device="cuda") * 0.1).half()
    for nv in (None, torch.tensor([max(B - 300, 77)], dtype=torch.int32, device="cuda")):
        rows = B if nv is None else min(B, (int(nv) + 127) // 128 * 128)
        res = []
        for fused in (False, True):
            net.train()
            net.zero_grad()
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.float16):
                if fused:
                    sigma, cin = net.forward_ngp_mid(x, d, level_major=True, n_valid=nv)
                else:
                    sigma, cin = _NgpMid.apply(net.forward_padded(x, level_major=True, n_valid=nv).contiguous(), d, nv)
            ((sigma[:rows] * gs[:rows]).sum() + (cin[:rows].float() * gc[:rows].float()).sum()).backward()
            res.append((sigma.detach()[:rows].clone(), cin.detach()[:rows].clone(), net.weights.grad.clone(),
                        x.grad[:, :rows].clone()))
        (s_a, c_a, w_a, x_a), (s_b, c_b, w_b, x_b) = res
        assert torch.equal(s_a, s_b) and torch.equal(w_a, w_b) and torch.equal(x_a, x_b)
        assert torch.equal(c_a[:, 16:], c_b[:, 16:])  # geometry features + zero pad: bit for bit
        # SH columns: the same sh_eval source inlined into two kernels; hipcc associates `K * T * cos` differently in the two
        # (fp32 results one ulp apart), which flips the fp16 rounding of a few values per 10^5
        sh_a, sh_b = c_a[:, :16].float(), c_b[:, :16].float()
        assert float((sh_a != sh_b).float().mean()) < 1e-4
        torch.testing.assert_close(sh_a, sh_b, rtol=1.0 / 1024, atol=2.0 ** -24)
        assert float(w_a.abs().max()) > 0 and float(sh_a.abs().max()) > 0
        net.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            s1, c1 = net.forward_ngp_mid(x0, d, level_major=True, n_valid=nv)
        assert torch.equal(s1[:rows], s_b) and torch.equal(c1[:rows], c_b)


@pytest.mark.parametrize("B", [128, 128 * 37, 128 * 2100])
def test_ngp_pair_matches_the_two_launches(hip, B):
    """FFMLP.forward_ngp_pair (s3d_ffmlp_ngp_pair_inference: density net + head + colour net + sigmoid in one launch, the
    inference loop's call) against forward_ngp_mid + forward_rgb: sigma bit for bit; rgb bit for bit except where the SH
    columns of the colour-net input differ by one fp16 ulp (a third inlined copy of sh_eval — see the density-head test)"""
    from ffmlp import FFMLP
    torch.manual_seed(11)
    sig, col = FFMLP(32, 16, 64, 2).cuda().eval(), FFMLP(32, 3, 64, 3).cuda().eval()
    assert sig.pair_supported(col)
    x = torch.randn(16, B, 2, device="cuda").half()  # level-major encoder output
    d = torch.nn.functional.normalize(torch.randn(B, 3, device="cuda"), dim=-1)
    for nv in (None, torch.tensor([max(B - 300, 77)], dtype=torch.int32, device="cuda")):
        rows = B if nv is None else min(B, (int(nv) + 127) // 128 * 128)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            s_a, cin = sig.forward_ngp_mid(x, d, level_major=True, n_valid=nv)
            rgb_a = col.forward_rgb(cin, n_valid=nv)
            s_b, rgb_b = sig.forward_ngp_pair(x, d, col, level_major=True, n_valid=nv)
        assert torch.equal(s_a[:rows], s_b[:rows])
        assert rgb_b.dtype == torch.float32 and rgb_b.shape == (B, 3)
        diff = (rgb_a[:rows] != rgb_b[:rows]).any(-1).float().mean()
        assert float(diff) < 2e-3, float(diff)
        torch.testing.assert_close(rgb_b[:rows], rgb_a[:rows], rtol=0, atol=4e-3)
        assert float(rgb_a[:rows].std()) > 1e-3
    x_row = x.permute(1, 0, 2).reshape(B, 32).contiguous()  # row-major inputs take the same kernel
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s_c, rgb_c = sig.forward_ngp_pair(x_row, d, col)
        s_d, rgb_d = sig.forward_ngp_pair(x, d, col, level_major=True)
    assert torch.equal(s_c, s_d) and torch.equal(rgb_c, rgb_d)


@pytest.mark.parametrize("B", [128 * 37, 128 * 2100])
def test_ngp_pair_training_step_matches_the_two_functions(hip, B):
    """FFMLP.forward_ngp_pair with gradients (one forward launch, the two fused backward kernels) against forward_ngp_mid +
    forward_rgb: outputs as in the inference test, every gradient to fp16 accuracy (the colour-net input's SH columns may differ
    by one fp16 ulp in a few rows per 10^5 between the kernels' inlined copies of sh_eval)"""
    from ffmlp import FFMLP
    torch.manual_seed(5)
    sig, col = FFMLP(32, 16, 64, 2).cuda(), FFMLP(32, 3, 64, 3).cuda()
    x0 = torch.randn(16, B, 2, device="cuda").half()
    d = torch.nn.functional.normalize(torch.randn(B, 3, device="cuda"), dim=-1)
    gs, gr = torch.randn(B, device="cuda") * 0.1, torch.randn(B, 3, device="cuda")
    nv = torch.tensor([B - 200], dtype=torch.int32, device="cuda")
    rows = min(B, (int(nv) + 127) // 128 * 128)
    res = []
    for pair in (False, True):
        sig.zero_grad(); col.zero_grad()
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            if pair:
                sigma, rgb = sig.forward_ngp_pair(x, d, col, level_major=True, n_valid=nv)
            else:
                sigma, cin = sig.forward_ngp_mid(x, d, level_major=True, n_valid=nv)
                rgb = col.forward_rgb(cin, n_valid=nv)
        ((sigma[:rows] * gs[:rows]).sum() + (rgb[:rows] * gr[:rows]).sum()).backward()
        res.append((sigma.detach()[:rows].clone(), rgb.detach()[:rows].clone(), sig.weights.grad.clone(), col.weights.grad.clone(),
                    x.grad[:, :rows].clone()))
    (s_a, r_a, ws_a, wc_a, x_a), (s_b, r_b, ws_b, wc_b, x_b) = res
    assert torch.equal(s_a, s_b)
    assert float((r_a != r_b).any(-1).float().mean()) < 2e-3
    for a, b in ((ws_a, ws_b), (wc_a, wc_b), (x_a.float(), x_b.float())):
        assert float(a.abs().max()) > 0
        torch.testing.assert_close(b, a, rtol=2e-2, atol=2e-3 * float(a.abs().max()))


@pytest.mark.parametrize("W,in_dim,n", [(16, 32, 2), (128, 32, 3), (256, 64, 2), (64, 128, 2)])
def test_ffmlp_module_other_widths_train_like_the_torch_twin(hip, W, in_dim, n):
    """`FFMLP` with the hidden widths ffmlp.cu:40-44 dispatches beside 32 / 64 (and an input wider than 64): forward, input
    gradient and weight gradient through the module's autograd path against the bias-free torch MLP on the same (fp16-rounded)
    weights — the reference's own comparison (testing/test_ffmlp.py:11-43)."""
    from ffmlp import FFMLP
    torch.manual_seed(W + n)
    net = FFMLP(in_dim, 3, W, n).cuda()
    x = (torch.randn(128 * 5 + 17, in_dim, device="cuda") * 0.5).requires_grad_(True)  # ragged batch: the module pads to 128
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    assert y.shape == (x.shape[0], 3)
    go = torch.randn_like(y.float()) * 0.1
    y.float().backward(go)
    w16 = net.weights.detach().half().float()
    mats = _split(w16, in_dim, W, n)
    xr = x.detach().half().float().requires_grad_(True)
    wr = [m.clone().requires_grad_(True) for m in mats]
    h = xr
    for i, m in enumerate(wr):
        h = h @ m.t()
        if i != n:
            h = torch.relu(h)
    ref = h[:, :3]
    ref.backward(go)
    torch.testing.assert_close(y.float(), ref.detach(), rtol=3e-2, atol=3e-2)
    assert (x.grad.float() - xr.grad).norm() / xr.grad.norm() < 3e-2
    gw_ref = torch.cat([m.grad.reshape(-1) for m in wr])
    gw = net.weights.grad.float()
    assert (gw - gw_ref).abs().max() / gw_ref.abs().max() < 3e-2
    # inference path (no_grad) gives the same outputs
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        yi = net(x.detach())
    torch.testing.assert_close(yi.float(), y.detach().float(), rtol=1e-3, atol=1e-3)
