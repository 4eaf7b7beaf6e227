"""CPU: the build's wrappers / renderer / network reproduce the reference's own Python (gridencoder/grid.py,
raymarching/raymarching.py, shencoder/sphere_harmonics.py, freqencoder/freq.py, ffmlp/ffmlp.py, nerf/renderer.py,
nerf/network.py) on identical seeded scenarios.  The expected values in tests/golden/wrappers.npz were produced by
RUNNING the reference's modules on top of the CPU oracle (oracle/gen_golden.py, section `wrappers`); here the build's
modules run on the same oracle, so every difference is a host-logic difference.  Bit-exact unless stated."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "wrappers.npz"))


def _seeded(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


GRID_CASES = {
    "hash": dict(input_dim=3, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=8, per_level_scale=2),
    "smooth": dict(input_dim=2, num_levels=3, level_dim=4, base_resolution=8, log2_hashmap_size=10, desired_resolution=64,
                   interpolation="smoothstep"),
    "tiled_ac": dict(input_dim=3, num_levels=3, level_dim=1, base_resolution=8, log2_hashmap_size=9, desired_resolution=32,
                     gridtype="tiled", align_corners=True),
}


@pytest.mark.parametrize("tag", list(GRID_CASES))
def test_grid_encoder_matches_reference_wrapper(oracle_wrappers, G, tag):
    enc = oracle_wrappers.gg.GridEncoder(**GRID_CASES[tag])
    assert np.array_equal(enc.offsets.numpy(), G[f"grid_{tag}_offsets"])
    assert float(enc.per_level_scale) == float(G[f"grid_{tag}_pls"])
    enc.embeddings.data.copy_(torch.from_numpy(G[f"grid_{tag}_emb"]))
    x = torch.from_numpy(G[f"grid_{tag}_x"]).requires_grad_(True)
    y = enc(x, bound=1)
    assert np.array_equal(y.detach().numpy(), G[f"grid_{tag}_y"])
    y.backward(torch.from_numpy(G[f"grid_{tag}_go"]))
    assert np.array_equal(enc.embeddings.grad.numpy(), G[f"grid_{tag}_gemb"])
    assert np.array_equal(x.grad.numpy(), G[f"grid_{tag}_gx"])


def test_lego_offsets_match_reference(oracle_wrappers, G):
    enc = oracle_wrappers.gg.GridEncoder(desired_resolution=2048)
    assert np.array_equal(enc.offsets.numpy(), G["grid_lego_offsets"])


def test_sh_freq_modules_match_reference(oracle_wrappers, G):
    d = torch.from_numpy(G["sh_d"]).requires_grad_(True)
    y = oracle_wrappers.sh.SHEncoder(degree=4)(d)
    assert np.array_equal(y.detach().numpy(), G["sh_y"])
    y.backward(_seeded(y.shape, 22, -1, 1))
    assert np.array_equal(d.grad.numpy(), G["sh_gd"])
    x = torch.from_numpy(G["freq_x"]).requires_grad_(True)
    yf = oracle_wrappers.fq.FreqEncoder(input_dim=3, degree=4)(x)
    assert np.array_equal(yf.detach().numpy(), G["freq_y"])
    yf.backward(_seeded(yf.shape, 24, -1, 1))
    assert np.array_equal(x.grad.numpy(), G["freq_gx"])


def test_ffmlp_module_matches_reference(oracle_wrappers, G):
    net = oracle_wrappers.ff.FFMLP(32, 3, 64, 3)
    assert net.num_parameters == int(G["ffmlp_num_parameters"]) and net.padded_output_dim == int(G["ffmlp_padded_out"])
    assert np.array_equal(net.weights.detach().numpy(), G["ffmlp_w"])  # same seed-42 init, same RNG consumption
    net.train()
    y = net(torch.from_numpy(G["ffmlp_x"]))
    assert np.array_equal(y.detach().float().numpy(), G["ffmlp_y"])


def test_march_wrapper_matches_reference(oracle_wrappers, G):
    from nerf import synthetic as syn
    rm = oracle_wrappers.rm
    _, bits = syn.lego_like_density_grid(seed=0)
    ro, rd = torch.from_numpy(G["march_ro"]), torch.from_numpy(G["march_rd"])
    nears, fars = rm.near_far_from_aabb(ro, rd, torch.tensor([-1.0, -1, -1, 1, 1, 1]), 0.2)
    assert np.array_equal(nears.numpy(), G["march_nears"]) and np.array_equal(fars.numpy(), G["march_fars"])
    counter = torch.zeros(2, dtype=torch.int32)
    torch.manual_seed(5)
    xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, 1.0, torch.from_numpy(bits), 1, 128, nears, fars, counter, -1, True, 128,
                                                   False, 0, 1024)
    assert np.array_equal(counter.numpy(), G["march_counter"]) and np.array_equal(rays.numpy(), G["march_rays"])
    assert list(xyzs.shape) == G["march_xyzs_shape"].tolist()
    assert np.array_equal(xyzs[:256].numpy(), G["march_xyzs_head"]) and np.array_equal(deltas[:256].numpy(), G["march_deltas_head"])
    assert np.array_equal(xyzs.double().sum(0).numpy(), G["march_xyzs_sum"])
    assert np.array_equal(deltas.double().sum(0).numpy(), G["march_deltas_sum"])


def test_renderer_matches_reference(oracle_wrappers, G):
    """update_extra_state (full sweep + partial update), run_cuda training branch and inference loop"""
    from nerf import renderer, synthetic as syn
    lo, hi = syn.lego_like_boxes(0)

    class Analytic(renderer.NeRFRenderer):
        def forward(self, x, dd):
            return syn.box_density(x, lo, hi, sigma=40.0), (x * 0.5 + 0.5).clamp(0, 1) * (0.5 + 0.5 * dd.abs())

        def density(self, x):
            return {"sigma": syn.box_density(x, lo, hi, sigma=40.0)}
    ro, rd = torch.from_numpy(G["march_ro"]), torch.from_numpy(G["march_rd"])
    R = Analytic(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, device_compaction=False)
    R.train()
    torch.manual_seed(7)
    R.update_extra_state()
    tr = R.run_cuda(ro[None], rd[None], bg_color=1, perturb=True, max_steps=1024)
    torch.manual_seed(8)
    R.update_extra_state()
    R.eval()
    ev = R.run_cuda(ro[None], rd[None], bg_color=1, perturb=False, max_steps=1024)
    assert np.array_equal(R.density_bitfield.numpy(), G["rend_bitfield"])
    assert R.mean_density == float(G["rend_mean_density"]) and R.mean_count == int(G["rend_mean_count"])
    assert R.iter_density == int(G["rend_iter_density"])
    assert np.array_equal(R.step_counter.numpy(), G["rend_step_counter"])
    assert np.array_equal(tr["image"][0].numpy(), G["rend_train_image"]) and np.array_equal(tr["depth"][0].numpy(), G["rend_train_depth"])
    assert np.array_equal(ev["image"][0].numpy(), G["rend_eval_image"]) and np.array_equal(ev["depth"][0].numpy(), G["rend_eval_depth"])


def test_sampling_path_without_occupancy_grid_matches_reference(oracle_wrappers, G):
    """`NeRFRenderer.run` (cuda_ray off: stratified + importance sampling, torch compositing; nerf/renderer.py:125-253) and
    `sample_pdf` (:12-46) — BASELINE configs[0]'s path — against the reference's own `run` executed on the same analytic
    scene: the 64x64 frame staged in chunks of 1,500 rays (eval), and a perturbed training batch (torch.rand, seed 11)."""
    from nerf import renderer, synthetic as syn
    lo, hi = syn.lego_like_boxes(0)

    class AnalyticRun(renderer.NeRFRenderer):
        def density(self, x):
            return {"sigma": syn.box_density(x, lo, hi, sigma=40.0)}

        def color(self, x, dd, mask=None, **kw):
            rgb = (x * 0.5 + 0.5).clamp(0, 1) * (0.5 + 0.5 * dd.abs())
            if mask is None:
                return rgb
            out = torch.zeros(mask.shape[0], 3, dtype=x.dtype)
            out[mask] = rgb[mask]
            return out
    ro, rd = torch.from_numpy(G["march_ro"]), torch.from_numpy(G["march_rd"])
    R = AnalyticRun(bound=1, cuda_ray=False, density_scale=1, min_near=0.2)
    R.eval()
    ev = R.render(ro[None], rd[None], staged=True, max_ray_batch=1500, bg_color=1, perturb=False, num_steps=64, upsample_steps=48)
    # (rays that miss the box have near = far = FLT_MAX: the reference's normalised depth is 0 / 0 = NaN there, and so is ours)
    assert np.array_equal(ev["image"][0].numpy(), G["run_eval_image"], equal_nan=True)
    assert np.array_equal(ev["depth"][0].numpy(), G["run_eval_depth"], equal_nan=True)
    R.train()
    torch.manual_seed(11)
    tr = R.render(ro[None, :1024], rd[None, :1024], bg_color=1, perturb=True, num_steps=64, upsample_steps=48)
    assert np.array_equal(tr["image"][0].numpy(), G["run_train_image"], equal_nan=True)
    assert np.array_equal(tr["depth"][0].numpy(), G["run_train_depth"], equal_nan=True)
    assert np.array_equal(tr["weights_sum"].numpy(), G["run_train_weights_sum"], equal_nan=True)
    assert float(torch.nan_to_num(ev["image"]).std()) > 0.05 and float(torch.nan_to_num(tr["weights_sum"]).max()) > 0.5  # (a real picture, not a blank)


def test_network_matches_reference(oracle_wrappers, G):
    """same parameter names/shapes as nerf/network.py (checkpoint compatibility) and the same forward"""
    from nerf import network
    net = network.NeRFNetwork(bound=1, cuda_ray=True, log2_hashmap_size=14)
    names = [k for k, _ in net.named_parameters()]
    assert names == G["net_param_names"].tolist()
    assert [str(tuple(p.shape)) for _, p in net.named_parameters()] == G["net_param_shapes"].tolist()
    for k, p in net.named_parameters():
        p.data.copy_(_seeded(p.shape, zlib.crc32(k.encode()) % 1000, -0.5, 0.5))
    sigma, color = net(torch.from_numpy(G["net_x"]), torch.from_numpy(G["net_d"]))
    # nn.Linear on CPU: identical op sequence -> bit-exact
    assert np.array_equal(sigma.detach().numpy(), G["net_sigma"]) and np.array_equal(color.detach().numpy(), G["net_color"])
