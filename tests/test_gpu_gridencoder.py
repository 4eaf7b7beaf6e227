"""GPU parity: grid encoder forward/backward/TV vs the CPU oracle through the C ABI.
Corner rows (hash / dense indexing) are integer work: BIT-EXACT.  Forward outputs follow the oracle's
operation order and are compared bit-exactly too; scatter-adds are order-dependent: tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _enc_meta(D=3, L=16, C=2, base=16, log2T=19, desired=2048, align_corners=False):
    pls = np.exp2(np.log2(desired / base) / (L - 1)) if L > 1 else 2.0
    offs, off = [], 0
    for i in range(L):
        res = int(np.ceil(base * pls ** i))
        n = min(2 ** log2T, (res if align_corners else res + 1) ** D)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return torch.tensor(offs, dtype=torch.int32), float(np.log2(pls)), off


def _inputs(B, D, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, D, generator=g)
    x[0] = 0.0
    x[1] = 1.0
    x[2, 0] = -0.01     # out of range -> zero output
    x[3, D - 1] = 1.001
    x[4] = 0.5
    return x.contiguous()


CASES = [
    # D, L, C, base, log2T, desired, gridtype, align, interp, dtype
    (3, 16, 2, 16, 19, 2048, 0, False, 0, torch.float32),   # Lego config, fp32
    (3, 16, 2, 16, 19, 2048, 0, False, 0, torch.float16),   # Lego config under -O
    (3, 4, 2, 4, 8, 32, 0, False, 0, torch.float32),        # reference test_hashgrid_grad.py config
    (2, 4, 4, 16, 12, 512, 0, False, 1, torch.float32),     # 2-D, smoothstep
    (3, 8, 1, 16, 15, 512, 1, True, 0, torch.float32),      # tiled, align_corners, C=1
    (3, 8, 8, 16, 14, 256, 0, False, 0, torch.float16),     # C=8 half
    (4, 4, 2, 8, 14, 64, 0, False, 0, torch.float32),       # 4-D
    (5, 2, 2, 4, 12, 8, 0, False, 0, torch.float32),        # 5-D
]


@pytest.mark.parametrize("D,L,C,base,log2T,desired,gridtype,align,interp,dtype", CASES)
def test_grid_forward_backward(oracle, hip, D, L, C, base, log2T, desired, gridtype, align, interp, dtype):
    offsets, S, total = _enc_meta(D, L, C, base, log2T, desired, align)
    B = 4096 + 37
    x = _inputs(B, D, seed=D * 100 + L)
    g = torch.Generator().manual_seed(1)
    emb = ((torch.rand(total, C, generator=g) * 2 - 1) * (1.0 if dtype == torch.float32 else 0.5)).to(dtype)
    want_jac = (dtype == torch.float32)
    out_c = torch.empty(L, B, C, dtype=dtype)
    jac_c = torch.empty(B, L * D * C, dtype=dtype) if want_jac else None
    cidx_c = torch.empty(B, L, 2 ** D, dtype=torch.int32)
    oracle.GridBackend.grid_encode_forward(x, emb, offsets, out_c, B, D, C, L, S, base, jac_c, gridtype, align, interp,
                                           corner_idx=cidx_c)
    xg, eg, og = x.cuda(), emb.cuda(), offsets.cuda()
    out_g = torch.empty(L, B, C, dtype=dtype, device="cuda")
    jac_g = torch.empty(B, L * D * C, dtype=dtype, device="cuda") if want_jac else None
    hip.GridBackend.grid_encode_forward(xg, eg, og, out_g, B, D, C, L, S, base, jac_g, gridtype, align, interp)
    cidx_g = torch.empty(B, L, 2 ** D, dtype=torch.int32, device="cuda")
    hip.GridBackend.grid_corner_indices(xg, og, cidx_g, B, D, C, L, S, base, gridtype, align)
    torch.cuda.synchronize()
    # 1. integer parity: every corner row of every (point, level)
    assert torch.equal(cidx_c, cidx_g.cpu()), "hash/dense indexing differs"
    # 2. outputs: same operation order as the oracle -> bit-exact
    a = out_c.view(torch.int16 if dtype == torch.float16 else torch.int32)
    b = out_g.cpu().view(torch.int16 if dtype == torch.float16 else torch.int32)
    assert torch.equal(a, b), f"max abs diff {(out_c.float() - out_g.cpu().float()).abs().max()}"
    assert (out_c[:, 2] == 0).all() and (out_c[:, 3] == 0).all()
    if want_jac:
        torch.testing.assert_close(jac_g.cpu(), jac_c, rtol=1e-6, atol=1e-7)

    # backward
    grad = torch.randn(L, B, C, generator=g).to(dtype)
    ge_c = torch.zeros(total, C, dtype=dtype)
    gi_c = torch.zeros(B, D, dtype=dtype) if want_jac else None
    oracle.GridBackend.grid_encode_backward(grad, x, emb, offsets, ge_c, B, D, C, L, S, base, jac_c, gi_c, gridtype, align, interp)
    if dtype == torch.float16 and C == 1:
        return
    # ---- path 1: direct global atomics (order-dependent float sums)
    hip.GridBackend.set_backward_path(1)
    ge_g = torch.zeros(total, C, dtype=dtype, device="cuda")
    gi_g = torch.zeros(B, D, dtype=dtype, device="cuda") if want_jac else None
    hip.GridBackend.grid_encode_backward(grad.cuda(), xg, eg, og, ge_g, B, D, C, L, S, base, jac_g, gi_g, gridtype, align, interp)
    torch.cuda.synchronize()
    # fp32 reference sum of the same (rounded) products for the half case
    ge_ref = ge_c
    if dtype == torch.float16:
        ge_ref = torch.zeros(total, C, dtype=torch.float32)
        oracle.GridBackend.grid_encode_backward(grad.float(), x, emb.float(), offsets, ge_ref, B, D, C, L, S, base, None, None,
                                                gridtype, align, interp)
    if dtype == torch.float32:
        torch.testing.assert_close(ge_g.cpu(), ge_c, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(gi_g.cpu(), gi_c, rtol=1e-5, atol=1e-5)
    else:
        # half2 atomics: the sum order is arbitrary in the reference too; half-precision random walk allowed
        err = (ge_g.cpu().float() - ge_ref).abs()
        big = ge_ref.abs() > 50
        assert (err[~big] <= 2e-3 * ge_ref.abs().clamp(min=1.0)[~big] * np.sqrt(8.0) * 8).all()
        assert (err[big] / ge_ref.abs()[big]).max() < 0.05 if big.any() else True
    # ---- path 2: binned — contributions partitioned by table slice, 64-bit fixed-point accumulation in LDS: exact
    # integer sums of the reference's products, deterministic
    runs = []
    try:
        for path in (2, 2):
            hip.GridBackend.set_backward_path(path)
            ge2 = torch.zeros(total, C, dtype=dtype, device="cuda")
            hip.GridBackend.grid_encode_backward(grad.cuda(), xg, eg, og, ge2, B, D, C, L, S, base, None, None, gridtype, align, interp)
            runs.append(ge2.cpu())
        torch.cuda.synchronize()
    finally:
        hip.GridBackend.set_backward_path(0)
    for r in runs[1:]:
        assert torch.equal(runs[0].view(torch.int16 if dtype == torch.float16 else torch.int32),
                           r.view(torch.int16 if dtype == torch.float16 else torch.int32)), "binned backward must be bit-reproducible"
    if dtype == torch.float32:
        torch.testing.assert_close(runs[0], ge_c, rtol=2e-5, atol=1e-5)  # the sequential fp32 oracle sum carries the rounding, not the exact integer sum
    else:
        # products are rounded to half as in the reference, the SUM is exact and rounded to half once
        torch.testing.assert_close(runs[0].float(), ge_ref, rtol=2e-3, atol=2e-3)
    ge_g = runs[0].cuda()
    # size-independent property: the table gradient sums to sum(grad) per level/channel (weights sum to 1)
    if dtype == torch.float32:
        valid = ((x >= 0) & (x <= 1)).all(-1)
        for l in (0, L - 1):
            want = grad[l][valid].double().sum(0)
            got = ge_g.cpu()[int(offsets[l]):int(offsets[l + 1])].double().sum(0)
            torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("D,L,C,base,log2T,desired,gridtype,interp,dtype,B", [
    (3, 16, 2, 16, 19, 2048, 0, 0, torch.float16, 70001),   # Lego config under -O, many chunks
    (3, 16, 2, 16, 19, 2048, 0, 0, torch.float32, 33000),
    (3, 6, 4, 16, 17, 512, 0, 1, torch.float16, 20000),
    (3, 4, 8, 16, 16, 128, 1, 0, torch.float32, 9000),      # tiled, C=8 (192-point chunks)
    (2, 8, 1, 16, 14, 1024, 0, 0, torch.float32, 50000),
    (4, 3, 2, 8, 15, 48, 0, 0, torch.float32, 12000),       # 4-D: 16 records per point and level
])
def test_grid_backward_binned_large(hip, D, L, C, base, log2T, desired, gridtype, interp, dtype, B):
    """Training-size batches take the binned path (count -> scatter -> LDS accumulate).  Checked with samples clustered
    in a few cells (heavy same-row traffic), zero-gradient tails and out-of-range points: reproducible bit for bit, the
    automatic choice IS the binned path, the fp32 result agrees with the direct-atomic kernel, and per level the
    table gradient sums to the sum of the in-range gradients (interpolation weights sum to one)."""
    offsets, S, total = _enc_meta(D, L, C, base, log2T, desired, False)
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, D, generator=g)
    x[: B // 4] = x[: B // 4] * 0.02 + 0.4            # a dense cluster: many identical rows
    x[B // 4: B // 4 + 100] = -0.5                    # out of range
    grad = (torch.randn(L, B, C, generator=g) * 1e-3).to(dtype)
    grad[:, B // 2: B // 2 + B // 8] = 0              # zero-gradient points
    grad[L - 1, : B // 3] = 0                         # zero on one level only
    xg, og, gg = x.cuda(), offsets.cuda(), grad.cuda()
    emb = torch.zeros(total, C, dtype=dtype, device="cuda")
    out = {}
    try:
        for tag, path in (("binned", 2), ("again", 2), ("auto", 0), ("atomics", 1)):
            hip.GridBackend.set_backward_path(path)
            ge = torch.zeros(total, C, dtype=dtype, device="cuda")
            hip.GridBackend.grid_encode_backward(gg, xg, emb, og, ge, B, D, C, L, S, base, None, None, gridtype, False, interp)
            out[tag] = ge.cpu()
    finally:
        hip.GridBackend.set_backward_path(0)
    it = torch.int16 if dtype == torch.float16 else torch.int32
    assert torch.equal(out["binned"].view(it), out["again"].view(it))
    assert torch.equal(out["binned"].view(it), out["auto"].view(it))
    assert out["binned"].float().abs().sum() > 0
    if dtype == torch.float32:
        torch.testing.assert_close(out["binned"], out["atomics"], rtol=1e-4, atol=1e-6)
    else:  # half atomics lose low bits on every add; the binned sum is exact and rounded once
        torch.testing.assert_close(out["binned"].float(), out["atomics"].float(), rtol=5e-2, atol=5e-3)
    valid = ((x >= 0) & (x <= 1)).all(-1)
    for l in (0, L - 1):
        want = grad[l][valid].double().sum(0)
        got = out["binned"][int(offsets[l]):int(offsets[l + 1])].double().sum(0)
        torch.testing.assert_close(got, want, rtol=2e-2 if dtype == torch.float16 else 1e-4, atol=2e-3)


def test_grid_backward_nonfinite_poisons_level(hip):
    offsets, S, total = _enc_meta(3, 4, 2, 16, 14, 64)
    B = 9000
    x = torch.rand(B, 3, generator=torch.Generator().manual_seed(5)).cuda()
    grad = torch.ones(4, B, 2, device="cuda")
    grad[2, 17, 1] = float("inf")
    emb = torch.zeros(total, 2, device="cuda")
    try:
        for path in (2,):
            hip.GridBackend.set_backward_path(path)
            ge = torch.zeros(total, 2, device="cuda")
            hip.GridBackend.grid_encode_backward(grad, x, emb, offsets.cuda(), ge, B, 3, 2, 4, S, 16, None, None, 0, False, 0)
            lv = ge[int(offsets[2]):int(offsets[3])]
            assert torch.isnan(lv).all(), f"path {path}: non-finite gradient must poison its level"
            assert torch.isfinite(ge[: int(offsets[2])]).all() and torch.isfinite(ge[int(offsets[3]):]).all()
    finally:
        hip.GridBackend.set_backward_path(0)


def test_grid_tv(oracle, hip):
    offsets, S, total = _enc_meta(3, 8, 2, 16, 15, 512)
    g = torch.Generator().manual_seed(2)
    emb = torch.rand(total, 2, generator=g)
    x = _inputs(5000, 3, seed=9)
    gr_c = torch.zeros(total, 2)
    oracle.GridBackend.grad_total_variation(x, emb, gr_c, offsets, 1e-3, 5000, 3, 2, 8, S, 16, 0, False)
    gr_g = torch.zeros(total, 2, device="cuda")
    hip.GridBackend.grad_total_variation(x.cuda(), emb.cuda(), gr_g, offsets.cuda(), 1e-3, 5000, 3, 2, 8, S, 16, 0, False)
    torch.testing.assert_close(gr_g.cpu(), gr_c, rtol=1e-4, atol=1e-8)


def test_grid_full_size_properties(hip):
    """B = 2^21 points on the Lego table: linearity in the table and partition of unity."""
    offsets, S, total = _enc_meta()
    B = 1 << 21
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, generator=g).cuda()
    og = offsets.cuda()
    run = lambda e: (lambda o: (hip.GridBackend.grid_encode_forward(x, e, og, o, B, 3, 2, 16, S, 16, None, 0, False, 0), o)[1])(
        torch.empty(16, B, 2, device="cuda"))
    ones = torch.ones(total, 2, device="cuda")
    o1 = run(ones)
    torch.testing.assert_close(o1, torch.ones_like(o1), rtol=0, atol=2e-6)  # weights sum to one
    e1 = torch.randn(total, 2, generator=g).cuda()
    e2 = torch.randn(total, 2, generator=g).cuda()
    torch.testing.assert_close(run(e1 + 2 * e2), run(e1) + 2 * run(e2), rtol=1e-4, atol=1e-5)


def test_grid_errors_raise(hip):
    x = torch.rand(8, 3).cuda()
    offs = torch.tensor([0, 8], dtype=torch.int32).cuda()
    emb = torch.rand(8, 3).cuda()
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        hip.GridBackend.grid_encode_forward(x, emb, offs, torch.empty(1, 8, 3, device="cuda"), 8, 3, 3, 1, 1.0, 16, None, 0, False, 0)
    with pytest.raises(RuntimeError, match="D must be 2, 3, 4, or 5"):
        hip.GridBackend.grid_encode_forward(torch.rand(8, 6).cuda(), torch.rand(8, 2).cuda(), offs,
                                            torch.empty(1, 8, 2, device="cuda"), 8, 6, 2, 1, 1.0, 16, None, 0, False, 0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_grid_backward_binned_training_size_vs_oracle(oracle, hip, dtype):
    """The binned backward on a training-shaped batch (Lego configuration, 40,000 samples marched along rays — consecutive
    samples share coarse cells, so the quad merging, the sorted bucket runs and the exact 2^24 fixed-point sum (fp16) are all
    exercised) against the CPU oracle: the oracle's fp32 sum of the same products for fp16 (the binned sum is exact and
    rounded once, the reference's half atomics are not), the oracle's own result for fp32."""
    from nerf import synthetic as syn
    D, L, C, base = 3, 16, 2, 16
    offsets, S, total = _enc_meta(D, L, C, base, 19, 2048, False)
    # ray-ordered points: 500 rays x 80 samples with the Lego step (2 sqrt(3) / 1024 in world units, half that in [0,1])
    g = torch.Generator().manual_seed(11)
    o = torch.rand(500, 3, generator=g) * 0.2 + 0.05
    d = torch.nn.functional.normalize(torch.rand(500, 3, generator=g) + 0.1, dim=-1)
    t = torch.arange(80).float() * (3.383e-3 / 2)
    x = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3).contiguous()
    B = x.shape[0]
    assert B == 40000 and float(x.max()) < 1.0
    grad = (torch.randn(L, B, C, generator=g) * 1e-2).to(dtype)
    grad[:, torch.rand(B, generator=g) < 0.25] = 0     # terminated samples
    emb = torch.zeros(total, C, dtype=dtype)
    ge_ref = torch.zeros(total, C, dtype=torch.float32)
    oracle.GridBackend.grid_encode_backward(grad.float(), x, emb.float(), offsets, ge_ref, B, D, C, L, S, base, None, None, 0, False, 0)
    hip.GridBackend.set_backward_path(2)
    try:
        ge = torch.zeros(total, C, dtype=dtype, device="cuda")
        hip.GridBackend.grid_encode_backward(grad.cuda(), x.cuda(), emb.cuda(), offsets.cuda(), ge, B, D, C, L, S, base, None, None, 0,
                                             False, 0)
    finally:
        hip.GridBackend.set_backward_path(0)
    got = ge.cpu().float()
    scale = float(ge_ref.abs().max())
    if dtype == torch.float32:
        torch.testing.assert_close(got, ge_ref, rtol=1e-4, atol=1e-5 * scale)
    else:
        # products rounded to fp16 (2^-11 relative each, as in the reference), merged runs rounded once: the error of a row
        # is a random walk of its contributions' roundings — bounded here by 2e-3 of the largest row plus 2e-3 relative
        torch.testing.assert_close(got, ge_ref, rtol=2e-3, atol=2e-3 * scale)
    assert int((got != 0).any(1).sum()) > 10000


def _ray_points(n_rays, n_steps, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.rand(n_rays, 3, generator=g) * 0.2 + 0.05
    d = torch.nn.functional.normalize(torch.rand(n_rays, 3, generator=g) + 0.1, dim=-1)
    t = torch.arange(n_steps).float() * (3.383e-3 / 2)
    return (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3).contiguous(), g


def test_grid_backward_fp16_large_values_and_clean_control_block(oracle, hip):
    """Third-generation binned backward, fp16: (1) records of 64 and more in magnitude take the general fixed-point conversion
    (the level's header word says so) and still match the oracle's fp32 sum; (2) a non-finite gradient poisons ITS level
    only and raises found_inf; (3) the self-cleaning control block is all-zero again after each of these calls, so the
    plain call that follows is bit-identical to the one before."""
    D, L, C, base = 3, 16, 2, 16
    offsets, S, total = _enc_meta(D, L, C, base, 19, 2048, False)
    x, g = _ray_points(400, 64, seed=21)
    B = x.shape[0]
    emb = torch.zeros(total, C, dtype=torch.float16)
    og, xg = offsets.cuda(), x.cuda()

    def run(grad, found=None):
        ge = torch.zeros(total, C, dtype=torch.float16, device="cuda")
        hip.GridBackend.grid_encode_backward(grad.cuda(), xg, emb.cuda(), og, ge, B, D, C, L, S, base, None, None, 0, False, 0,
                                             found_inf=found)
        return ge.cpu()

    def ref(grad):
        r = torch.zeros(total, C, dtype=torch.float32)
        oracle.GridBackend.grid_encode_backward(grad.float(), x, emb.float(), offsets, r, B, D, C, L, S, base, None, None, 0, False, 0)
        return r

    small = (torch.randn(L, B, C, generator=g) * 1e-2).half()
    small[:, (torch.arange(B) % 64) >= 50] = 0            # ray tails behind the termination
    a0 = run(small)
    r0 = ref(small)
    torch.testing.assert_close(a0.float(), r0, rtol=2e-3, atol=2e-3 * float(r0.abs().max()))
    # (1) large records on some levels: |v| up to ~400 (level 3 and 9), the rest small
    big = small.clone()
    big[3] = (torch.randn(B, C, generator=g) * 100).half()
    big[9] = (torch.randn(B, C, generator=g) * 100).half()
    big[:, (torch.arange(B) % 64) >= 50] = 0
    a1 = run(big)
    r1 = ref(big)
    for l in range(L):
        sl = slice(int(offsets[l]), int(offsets[l + 1]))
        torch.testing.assert_close(a1[sl].float(), r1[sl], rtol=2e-3, atol=2e-3 * float(r1[sl].abs().max()))
    assert torch.equal(run(small).view(torch.int16), a0.view(torch.int16)), "control block not clean after the large-value call"
    # (2) non-finite gradient on level 5
    bad = small.clone()
    bad[5, 1234, 1] = float("inf")
    found = torch.zeros(1, device="cuda")
    a2 = run(bad, found)
    assert float(found) == 1.0
    lv = a2[int(offsets[5]):int(offsets[6])]
    assert torch.isnan(lv.float()).all(), "non-finite gradient must poison its level"
    keep = torch.ones(total, dtype=torch.bool)
    keep[int(offsets[5]):int(offsets[6])] = False
    assert torch.equal(a2[keep].view(torch.int16), a0[keep].view(torch.int16)), "other levels must be untouched by the poison"
    # (3) ... and the next call is clean again
    assert torch.equal(run(small).view(torch.int16), a0.view(torch.int16)), "control block not clean after the poisoned call"
    # a merged run whose sum leaves binary16 (64 samples of one cell, 2000 each): poisoned as well
    xs = torch.full((8192, 3), 0.31)
    gs = torch.zeros(L, 8192, C, dtype=torch.float16)
    gs[0] = 2000.0
    ge = torch.zeros(total, C, dtype=torch.float16, device="cuda")
    found.zero_()
    hip.GridBackend.grid_encode_backward(gs.cuda(), xs.cuda(), emb.cuda(), og, ge, 8192, D, C, L, S, base, None, None, 0, False, 0,
                                         found_inf=found)
    assert float(found) == 1.0 and torch.isnan(ge[: int(offsets[1])].float()).all()
    assert torch.equal(run(small).view(torch.int16), a0.view(torch.int16))


def test_grid_backward_binned_inside_graph_capture(hip):
    """The binned backward captured into a HIP graph (the control block comes from the binding, allocated outside the
    capture): replays are bit-identical to the eager call."""
    D, L, C, base = 3, 16, 2, 16
    offsets, S, total = _enc_meta(D, L, C, base, 19, 2048, False)
    x, g = _ray_points(300, 64, seed=5)
    B = x.shape[0]
    grad = (torch.randn(L, B, C, generator=g) * 1e-2).half().cuda()
    xg, og = x.cuda(), offsets.cuda()
    emb = torch.zeros(total, C, dtype=torch.float16, device="cuda")
    ge = torch.zeros(total, C, dtype=torch.float16, device="cuda")

    def call():
        hip.GridBackend.grid_encode_backward(grad, xg, emb, og, ge, B, D, C, L, S, base, None, None, 0, False, 0)
    call()
    want = ge.clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ge.zero_()
        call()
    torch.cuda.current_stream().wait_stream(st)
    graph = torch.cuda.CUDAGraph()
    ge.zero_()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        call()
    for _ in range(3):
        ge.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(ge.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("dtype,C,gridtype", [(torch.float16, 2, 0), (torch.float32, 1, 0), (torch.float16, 2, 1)])
def test_grid_backward_plane_of_group_boundary_cells(oracle, hip, dtype, C, gridtype):
    """Binned backward on a batch built against the slice interleave: half of the points sit in x cells that are 31 mod 32 on
    one level — the two x-corners of every corner pair then fall into DIFFERENT 32-row groups, i.e. different table slices —
    in distinct cells (nothing merges), on hashed and on tiled (wrapping, non power-of-two) tables; the other half random.
    All: equal to the oracle's sums; fp16: the binned sums are exact, i.e. bit-identical from run to run."""
    D, L, base = 3, 16, 16
    offsets, S, total = _enc_meta(D, L, C, base, 19, 2048, False)
    g = torch.Generator().manual_seed(5)
    B = 24000
    lvl = 9
    scale = float(2.0 ** (lvl * S) * base - 1.0)
    res = int(np.ceil(scale)) + 1
    cells = torch.arange(31, res - 1, 32)
    cx = cells[torch.randint(0, len(cells), (B,), generator=g)].float()
    x = torch.rand(B, D, generator=g)
    x[:, 0] = (cx + 0.25 - 0.5) / scale            # pos = x * scale + 0.5 -> cell cx, fraction 0.25
    x[B // 2:] = torch.rand(B - B // 2, D, generator=g)
    assert float(x.min()) >= 0 and float(x.max()) <= 1
    pg = torch.floor(x[: B // 2, 0] * scale + 0.5).long()
    assert bool((pg % 32 == 31).all())
    grad = (torch.randn(L, B, C, generator=g) * 1e-2).to(dtype)
    emb = torch.zeros(total, C, dtype=dtype)
    ge_ref = torch.zeros(total, C, dtype=torch.float32)
    oracle.GridBackend.grid_encode_backward(grad.float(), x, emb.float(), offsets, ge_ref, B, D, C, L, S, base, None, None, gridtype,
                                            False, 0)
    out = {}
    try:
        for tag, path in ((2, 2), ("again", 2), (1, 1)):
            hip.GridBackend.set_backward_path(path)
            ge = torch.zeros(total, C, dtype=dtype, device="cuda")
            hip.GridBackend.grid_encode_backward(grad.cuda(), x.cuda(), emb.cuda(), offsets.cuda(), ge, B, D, C, L, S, base, None, None,
                                                 gridtype, False, 0)
            out[tag] = ge.cpu()
    finally:
        hip.GridBackend.set_backward_path(0)
    if dtype == torch.float16:
        # (binned: one rounding to binary16 after an exact sum; atomics: one rounding per add, arrival order)
        torch.testing.assert_close(out[2].float(), ge_ref, rtol=2e-3, atol=2e-3 * float(ge_ref.abs().max()))
        torch.testing.assert_close(out[1].float(), ge_ref, rtol=2e-2, atol=2e-2 * float(ge_ref.abs().max()))
        assert torch.equal(out[2].view(torch.int16), out["again"].view(torch.int16))  # exact sums: the same bits every time
    else:
        torch.testing.assert_close(out[2], ge_ref, rtol=1e-4, atol=1e-5 * float(ge_ref.abs().max()))


@pytest.mark.parametrize("pair", [False, True])
def test_forward_of_a_padded_batch_with_a_device_side_row_count(hip, pair):
    """A padded batch (extent of more than 2,048 chunks, `n_valid` rows filled — the teacher's proxy render of the Seal step:
    N x max_steps rows, 2e5 of them real) is walked by a fixed number of workgroups (k_grid_forward_pair<..., STRIDED>): the
    filled rows equal, bit for bit, what a launch over exactly those rows writes; rows behind the count (rounded up to 128, as
    everywhere: valid_rows()) are not touched."""
    torch.manual_seed(7)
    offsets, S, total = _enc_meta()
    D, L, C, base = 3, 16, 2, 16
    B, nv = 3 * 2048 * 256 + 1000, 200_000 + 77          # ragged extent, ragged count
    G = hip.GridBackend
    x = torch.rand(B, D, device="cuda")
    x[5] = 1.5                                            # an out-of-range row inside the filled part
    emb = [(torch.rand(total, C, device="cuda") * 2 - 1).half() for _ in range(2)]
    offs = offsets.cuda()
    n_valid = torch.tensor([nv], dtype=torch.int32, device="cuda")
    rows = (nv + 255) // 256 * 256                        # a launch over the filled chunks only
    out = [torch.full((L, B, C), 7.0, device="cuda", dtype=torch.half) for _ in range(2)]
    ref = [torch.empty(L, rows, C, device="cuda", dtype=torch.half) for _ in range(2)]
    if pair:
        G.grid_encode_forward_pair(x, emb[0], emb[1], offs, out[0], out[1], B, D, C, L, S, base, 0, False, 0, 0.0, n_valid)
        G.grid_encode_forward_pair(x[:rows].contiguous(), emb[0], emb[1], offs, ref[0], ref[1], rows, D, C, L, S, base, 0, False, 0)
    else:
        G.grid_encode_forward(x, emb[0], offs, out[0], B, D, C, L, S, base, None, 0, False, 0, 0.0, n_valid)
        G.grid_encode_forward(x[:rows].contiguous(), emb[0], offs, ref[0], rows, D, C, L, S, base, None, 0, False, 0)
    for k in range(2 if pair else 1):
        assert torch.equal(out[k][:, :nv], ref[k][:, :nv])
        assert float(out[k][:, :nv].abs().max()) > 0 and bool((out[k][:, 5] == 0).all())
        assert bool((out[k][:, (nv + 127) // 128 * 128:] == 7.0).all())  # (the count is honoured in units of 128 rows: valid_rows())
