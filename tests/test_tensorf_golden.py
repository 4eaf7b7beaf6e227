"""CPU: the build's TensoRF backbone and trainer against the REFERENCE's, executed (tests/golden/tensorf.npz from
`oracle/gen_golden.py tensorf`: tensoRF/network.py NeRFNetwork forward / density / density_loss / get_params /
shrink_model / upsample_model, tensoRF/utils.py Trainer.train_step incl. the L1 term, SealNeRF/trainer.py freeze_mlp +
pretrain_step on the TensoRF backbone).  The native kernels under the drop-in packages are the CPU oracle's here; the factor
sampling is torch's grid_sample on both sides."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN

NET = dict(resolution=[24, 28, 32], sigma_rank=[4, 5, 6], color_rank=[6, 7, 8], bound=1, cuda_ray=True, density_scale=1,
           min_near=0.2, density_thresh=10)


def _seeded(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


@pytest.fixture(scope="module")
def T():
    return np.load(os.path.join(GOLDEN, "tensorf.npz"))


def _net():
    from nerf import synthetic as syn
    from tensoRF import network as trf
    net = trf.NeRFNetwork(**NET)
    for k, p in net.named_parameters():
        p.data.copy_(_seeded(p.shape, zlib.crc32(k.encode()) % 1000, -0.5, 0.5))
    dens, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(dens))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    return net, dens


def _check_grads(net, T, prefix, rtol=1e-6):
    for k, p in net.named_parameters():
        key = f"{prefix}_{k.replace('.', '_')}"
        g = p.grad.detach().reshape(-1)
        assert abs(float(g.double().norm()) - float(T[key + "_norm"])) <= rtol * max(float(T[key + "_norm"]), 1e-12), k
        if key in T.files:
            np.testing.assert_allclose(g.numpy(), T[key], rtol=1e-5, atol=rtol * float(np.abs(T[key]).max()), err_msg=k)
        else:
            np.testing.assert_allclose(g[torch.from_numpy(T[key + "_at"])].numpy(), T[key + "_at_values"], rtol=1e-5,
                                       atol=rtol * float(np.abs(T[key + "_at_values"]).max()), err_msg=k)


def test_parameters_groups_forward_and_density_loss(oracle_wrappers, T):
    net, _ = _net()
    assert [k for k, _ in net.named_parameters()] == T["param_names"].tolist()
    assert [str(tuple(p.shape)) for _, p in net.named_parameters()] == T["param_shapes"].tolist()
    names = {id(p): k for k, p in net.named_parameters()}
    groups = net.get_params(2e-2, 1e-3)
    assert [g["lr"] for g in groups] == T["group_lrs"].tolist()
    assert [",".join(names[id(p)] for p in g["params"]) for g in groups] == T["group_members"].tolist()
    x, d = torch.from_numpy(T["fw_x"]), torch.from_numpy(T["fw_d"])
    with torch.no_grad():
        sg, cl = net(x, d)
        np.testing.assert_allclose(sg.numpy(), T["fw_sigma"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cl.numpy(), T["fw_color"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(net.density(x)["sigma"].numpy(), T["fw_density"], rtol=1e-6, atol=1e-7)
        assert abs(float(net.density_loss()) - float(T["density_loss"])) <= 1e-6 * float(T["density_loss"])


def test_train_step_with_the_l1_term_matches_the_reference_trainer(oracle_wrappers, T, monkeypatch):
    from tensoRF.utils import Trainer
    net, _ = _net()
    net.mean_count = 32768
    tr = Trainer(net, lr0=2e-2, lr1=1e-3, l1_reg_weight=float(T["ts_l1_weight"]), fp16=False, update_extra_interval=10 ** 9)
    assert [g["lr"] for g in tr.optimizer.param_groups] == T["group_lrs"].tolist()
    tr.global_step = 1
    net.train()
    torch.manual_seed(5)
    ro, rd, gt = (torch.from_numpy(T[k]) for k in ("ts_rays_o", "ts_rays_d", "ts_images"))
    seen = {}
    monkeypatch.setattr(tr, "_reduce_and_step", lambda: seen.update({k: p.grad.clone() for k, p in net.named_parameters()}))
    loss = tr.train_step(ro[0], rd[0], gt[0])
    assert abs(float(loss) - float(T["ts_loss"])) <= 1e-6 * float(T["ts_loss"])
    assert np.array_equal(net.step_counter[0].numpy(), T["ts_counter"])
    for k, p in net.named_parameters():
        p.grad = seen[k]
    _check_grads(net, T, "ts_grad")


def test_seal_pretraining_freezes_nothing_on_the_tensorf_backbone(oracle_wrappers, T):
    from sealnerf import SealTrainer
    net, _ = _net()
    teacher, _ = _net()
    tr = SealTrainer(net, teacher, lr=2e-2, fp16=False, update_extra_interval=10 ** 9)
    tr.freeze_mlp(True)
    assert [k for k, p in net.named_parameters() if not p.requires_grad] == T["pt_frozen"].tolist() == []
    pts, dirs, gs, gc = (torch.from_numpy(T[k]) for k in ("pt_points", "pt_dirs", "pt_sigma", "pt_color"))
    net.zero_grad()
    loss = tr.pretrain_loss(pts, dirs, gs, gc)
    loss.backward()
    assert abs(float(loss) - float(T["pt_loss"])) <= 1e-6 * float(T["pt_loss"])
    _check_grads(net, T, "pt_grad")
    tr.freeze_mlp(False)


def test_shrink_and_upsample_follow_the_reference(oracle_wrappers, T):
    net, dens = _net()
    net.mean_density = 5.0
    net.shrink_model()
    np.testing.assert_allclose(net.aabb_train.numpy(), T["shrink_aabb"], rtol=0, atol=1e-7)
    assert [str(tuple(p.shape)) for _, p in net.named_parameters()] == T["shrink_shapes"].tolist()
    net.upsample_model([30, 26, 22])
    assert [str(tuple(p.shape)) for _, p in net.named_parameters()] == T["up_shapes"].tolist()
    with torch.no_grad():
        sg, cl = net(torch.from_numpy(T["up_x"]), torch.from_numpy(T["fw_d"][:200]))
    np.testing.assert_allclose(sg.numpy(), T["up_sigma"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(cl.numpy(), T["up_color"], rtol=1e-5, atol=1e-7)
