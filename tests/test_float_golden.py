"""The compositing kernels against the REFERENCE TEXT at north_star's tolerance (tests/golden/float_kernels.npz).

`oracle/gen_golden.py float` transliterates `kernel_composite_rays_train_forward` / `_backward` (raymarching.cu:501-684) and
`kernel_composite_rays` (:821-900) statement by statement and runs them thread by thread with numpy float32 scalars (`__expf`
as float32 exp, no fused multiply-adds: what nvcc contracts cannot be observed here and is far below the tolerance).  The CPU
oracle must agree within 1e-4 relative (north_star: "within 1e-4 rel on composited RGB / sigma") and exactly in every integer
(the kill pattern of the inference loop, the zeros of empty / overflowing rays); the HIP kernels: tests/test_gpu_golden.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

RTOL = 1e-4


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "float_kernels.npz"))


def close(got, want, what):
    want = np.asarray(want, np.float64)
    np.testing.assert_allclose(np.asarray(got, np.float64), want, rtol=RTOL, atol=RTOL * 1e-2 * max(float(np.abs(want).max()), 1e-30),
                               err_msg=what)


def train_compositing(R, G, dev="cpu"):
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    sig, rgb, dl, rays = t("ct_sigmas"), t("ct_rgbs"), t("ct_deltas"), t("ct_rays")
    M, N, Tt = int(G["ct_M"]), rays.shape[0], float(G["ct_T_thresh"])
    ws, dp, im = (torch.full((N,), -7.0, device=dev), torch.full((N,), -7.0, device=dev), torch.full((N, 3), -7.0, device=dev))
    R.composite_rays_train_forward(sig, rgb, dl, rays, M, N, Tt, ws, dp, im)
    g_sig, g_rgb = torch.zeros_like(sig), torch.zeros_like(rgb)
    # (the backward reads the forward's OWN results, as autograd hands them over)
    R.composite_rays_train_backward(t("ct_grad_weights_sum"), t("ct_grad_image"), sig, rgb, dl, rays, ws, im, M, N, Tt, g_sig, g_rgb)
    return [v.cpu().numpy() for v in (ws, dp, im, g_sig, g_rgb)]


def check_train(got, G):
    ws, dp, im, g_sig, g_rgb = got
    rays, M = G["ct_rays"], int(G["ct_M"])
    empty = (rays[:, 2] == 0) | (rays[:, 1] + rays[:, 2] > M)
    assert empty.sum() >= 10 and (rays[:, 1] + rays[:, 2] > M).any()
    for a in (ws, dp, im):
        assert (a[rays[empty, 0]] == 0).all()                     # empty rays and the ray past the buffer's end: exact zeros
    close(ws, G["ct_weights_sum"], "weights_sum")
    close(dp, G["ct_depth"], "depth")
    close(im, G["ct_image"], "image")
    close(g_rgb, G["ct_grad_rgbs"], "grad_rgbs")
    close(g_sig, G["ct_grad_sigmas"], "grad_sigmas")
    assert np.array_equal(g_sig == 0, G["ct_grad_sigmas"] == 0)  # samples behind a termination / outside every span stay untouched


def inference_compositing(R, G, dev="cpu"):
    n_step, Tt, NR = int(G["ci_n_step"]), float(G["ci_T_thresh"]), int(G["ci_NR"])
    rays_t = torch.from_numpy(G["ci_rays_t_init"].copy()).to(dev)
    ws, dp, im = torch.zeros(NR, device=dev), torch.zeros(NR, device=dev), torch.zeros(NR, 3, device=dev)
    out = []
    for it in range(3):
        alive = torch.from_numpy(G[f"ci{it}_alive_in"].copy()).to(dev)
        alive = alive[alive >= 0].contiguous()
        n_alive = alive.shape[0]
        s, c, d = (torch.from_numpy(G[f"ci{it}_{k}"]).to(dev) for k in ("sigmas", "rgbs", "deltas"))
        R.composite_rays(n_alive, n_step, Tt, alive, rays_t, s, c, d, ws, dp, im)
        out.append([v.cpu().numpy().copy() for v in (alive, rays_t, ws, dp, im)])
    return out


def check_inference(out, G):
    killed = 0
    for it, (alive, rays_t, ws, dp, im) in enumerate(out):
        want_alive = G[f"ci{it}_alive_out"]
        assert np.array_equal(alive, want_alive[:alive.shape[0]] if want_alive.shape[0] != alive.shape[0] else want_alive), it
        killed += int((alive < 0).sum())
        close(rays_t, G[f"ci{it}_rays_t"], "rays_t")
        close(ws, G[f"ci{it}_weights_sum"], "weights_sum")
        close(dp, G[f"ci{it}_depth"], "depth")
        close(im, G[f"ci{it}_image"], "image")
    assert killed > 50


def test_training_compositing_forward_and_backward(oracle, G):
    check_train(train_compositing(oracle.RaymarchingBackend, G), G)


def test_inference_compositing_three_iterations(oracle, G):
    check_inference(inference_compositing(oracle.RaymarchingBackend, G), G)
