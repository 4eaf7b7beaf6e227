"""CPU: the build's Seal fine-tuning / pretraining losses, network and renderer on the CPU oracle reproduce
tests/golden/trainstep.npz — the REFERENCE's own `Trainer.train_step` (nerf/utils.py:436-537), `pretrain_step` +
`freeze_mlp` (SealNeRF/trainer.py:455-488), `NeRFNetwork.render` and `PSNRMeter` (nerf/utils.py:215-240) executed on the
same oracle (oracle/gen_golden.py train).  Same torch CPU ops on both sides: loss and gradients agree to the last bits
(tolerance 1e-6 relative for reduction-order freedom inside torch)."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN

NET = dict(bound=1, cuda_ray=True, log2_hashmap_size=14, density_scale=1, min_near=0.2, density_thresh=10)


def _seeded(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


@pytest.fixture(scope="module")
def T():
    return np.load(os.path.join(GOLDEN, "trainstep.npz"))


def _student():
    from nerf import network, synthetic as syn
    net = network.NeRFNetwork(**NET)
    for k, p in net.named_parameters():
        p.data.copy_(_seeded(p.shape, zlib.crc32(k.encode()) % 1000, -0.5, 0.5))
    dens, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(dens))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    return net


def _check_grads(net, T, prefix, frozen=(), rtol=1e-6):
    for k, p in net.named_parameters():
        key = f"{prefix}_{k.replace('.', '_')}"
        if k in frozen:
            assert key + "_none" in T.files and p.grad is None, k
            continue
        g = p.grad.detach()
        assert abs(float(g.double().norm()) - float(T[key + "_norm"])) <= rtol * float(T[key + "_norm"]), k
        if key in T.files:
            np.testing.assert_allclose(g.numpy(), T[key], rtol=1e-5, atol=rtol * float(np.abs(T[key]).max()), err_msg=k)
        else:
            np.testing.assert_allclose(g[torch.from_numpy(T[key + "_rows"])].numpy(), T[key + "_at_rows"], rtol=1e-5,
                                       atol=rtol * float(np.abs(T[key + "_at_rows"]).max()), err_msg=k)


def test_finetune_loss_matches_reference_train_step(oracle_wrappers, T):
    from sealnerf import SealTrainer
    net = _student()
    net.mean_count = int(T["ts_mean_count"])
    tr = SealTrainer(net, net, lr=1e-2, fp16=False)
    net.train()
    ro, rd = torch.from_numpy(T["ts_rays_o"]), torch.from_numpy(T["ts_rays_d"])
    torch.manual_seed(5)
    loss, out = tr.finetune_loss(ro, rd, torch.from_numpy(T["ts_images"]), torch.from_numpy(T["ts_depths"]), bg_color=1)
    assert np.array_equal(net.step_counter[0].numpy(), T["ts_counter"])
    assert abs(float(loss) - float(T["ts_loss"])) <= 1e-6 * float(T["ts_loss"])
    np.testing.assert_allclose(out["image"].detach().numpy(), T["ts_pred"], rtol=1e-6, atol=1e-7)
    net.zero_grad()
    loss.backward()
    _check_grads(net, T, "ts_grad")
    net.local_step = 0
    torch.manual_seed(5)
    loss_rgb, _ = tr.finetune_loss(ro, rd, torch.from_numpy(T["ts_images"]), None, bg_color=1)
    assert abs(float(loss_rgb) - float(T["ts_loss_rgb_only"])) <= 1e-6 * float(T["ts_loss_rgb_only"])


def test_pretrain_loss_matches_reference_pretrain_step(oracle_wrappers, T):
    from sealnerf import SealTrainer
    net = _student()
    tr = SealTrainer(net, net, lr=1e-2, fp16=False)
    net.train()
    tr.freeze_mlp(True)
    frozen = [k for k, p in net.named_parameters() if not p.requires_grad]
    assert frozen == T["pt_frozen"].tolist()
    net.zero_grad()
    loss = tr.pretrain_loss(torch.from_numpy(T["pt_points"]), torch.from_numpy(T["pt_dirs"]), torch.from_numpy(T["pt_sigma"]),
                            torch.from_numpy(T["pt_color"]))
    assert abs(float(loss) - float(T["pt_loss"])) <= 1e-6 * float(T["pt_loss"])
    loss.backward()
    _check_grads(net, T, "pt_grad", frozen=frozen)


def test_eval_render_and_psnr_match_reference(oracle_wrappers, T):
    from nerf.trainer import psnr
    net = _student()
    net.eval()
    net.device_compaction = False
    with torch.no_grad():
        ev = net.render(torch.from_numpy(T["ev_rays_o"]), torch.from_numpy(T["ev_rays_d"]), bg_color=1, perturb=False, max_steps=1024,
                        T_thresh=1e-4, dt_gamma=0)
    assert np.array_equal(ev["image"].numpy(), T["ev_image"]) and np.array_equal(ev["depth"].numpy(), T["ev_depth"])
    assert abs(psnr(ev["image"], torch.from_numpy(T["ev_truth"])) - float(T["ev_psnr"])) < 1e-5  # PSNRMeter formula
