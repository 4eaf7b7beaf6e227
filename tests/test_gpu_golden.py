"""GPU: the drop-in packages on libseal3d_hip.so against the fixtures the REFERENCE's own Python produced
(oracle/gen_golden.py; tests/golden/wrappers.npz, trainstep.npz, seal_bbox.npz).

The expected values were computed by the reference's modules (gridencoder/grid.py, raymarching/raymarching.py,
shencoder, freqencoder, ffmlp, nerf/renderer.py, nerf/network.py, nerf/utils.py Trainer.train_step / PSNRMeter,
SealNeRF/trainer.py pretrain_step, SealNeRF/seal_utils.py map_to_origin) running on the CPU oracle; here the BUILD's
modules run on the HIP library.  Bars (north_star): integers (rays, counters, compaction, bitfield) bit-exact; fp32 hash
features bit-exact; composited RGB / depth / sigma gradients within 1e-4 relative; PSNR within 0.1 dB.

Random numbers: the fixtures were drawn from torch's CPU generator.  The GPU modules draw from the device generator, so the
tests route the modules' `torch.rand*` calls through the CPU generator (same shapes, same order) and move the result to the
device — the arithmetic under test is unchanged."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


class _CpuRandom:
    """stand-in for the `torch` name inside a module: rand / rand_like / randint are drawn on the CPU generator"""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def _dev(kw):
        dev = kw.pop("device", None)
        return dev

    def rand(self, *a, **kw):
        dev = self._dev(kw)
        return torch.rand(*a, **kw).to(dev) if dev is not None else torch.rand(*a, **kw)

    def rand_like(self, t, **kw):
        return torch.rand(t.shape, dtype=t.dtype).to(t.device)

    def randint(self, *a, **kw):
        dev = self._dev(kw)
        return torch.randint(*a, **kw).to(dev) if dev is not None else torch.randint(*a, **kw)


@pytest.fixture()
def cpu_random(monkeypatch):
    import raymarching.raymarching as rm
    import nerf.renderer as rend
    proxy = _CpuRandom()
    monkeypatch.setattr(rm, "torch", proxy)
    monkeypatch.setattr(rend, "torch", proxy)
    return proxy


def _seeded(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def _close(a, b, rtol=1e-4, atol=1e-6, what=""):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol, err_msg=what)


def _relmax(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ------------------------------------------------------------------------------------------------ wrappers.npz
GRID_CASES = {
    "hash": dict(input_dim=3, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=8, per_level_scale=2),
    "smooth": dict(input_dim=2, num_levels=3, level_dim=4, base_resolution=8, log2_hashmap_size=10, desired_resolution=64,
                   interpolation="smoothstep"),
    "tiled_ac": dict(input_dim=3, num_levels=3, level_dim=1, base_resolution=8, log2_hashmap_size=9, desired_resolution=32,
                     gridtype="tiled", align_corners=True),
}


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "wrappers.npz"))


@pytest.mark.parametrize("tag", list(GRID_CASES))
def test_grid_encoder_module_vs_reference_fixture(hip, G, tag):
    from gridencoder import GridEncoder
    enc = GridEncoder(**GRID_CASES[tag]).cuda()
    assert np.array_equal(enc.offsets.cpu().numpy(), G[f"grid_{tag}_offsets"])
    enc.embeddings.data.copy_(torch.from_numpy(G[f"grid_{tag}_emb"]))
    x = torch.from_numpy(G[f"grid_{tag}_x"]).cuda().requires_grad_(True)
    y = enc(x, bound=1)
    assert np.array_equal(y.detach().cpu().numpy(), G[f"grid_{tag}_y"]), "fp32 hash-grid features must be bit-exact"
    y.backward(torch.from_numpy(G[f"grid_{tag}_go"]).cuda())
    # the table gradient is a sum of up to 257 * 2^D float products per row in a different order: 1e-4 of the largest term
    _close(enc.embeddings.grad, G[f"grid_{tag}_gemb"], rtol=1e-4, atol=1e-4 * float(np.abs(G[f"grid_{tag}_gemb"]).max()), what="grad_embeddings")
    _close(x.grad, G[f"grid_{tag}_gx"], rtol=1e-4, atol=1e-4 * float(np.abs(G[f"grid_{tag}_gx"]).max()), what="grad_inputs")


def test_sh_freq_ffmlp_modules_vs_reference_fixture(hip, G):
    from shencoder import SHEncoder
    from freqencoder import FreqEncoder
    from ffmlp import FFMLP
    d = torch.from_numpy(G["sh_d"]).cuda().requires_grad_(True)
    y = SHEncoder(degree=4)(d)
    _close(y, G["sh_y"], rtol=2e-5, atol=1e-5)
    y.backward(_seeded(y.shape, 22, -1, 1).cuda())
    _close(d.grad, G["sh_gd"], rtol=1e-4, atol=1e-4)
    x = torch.from_numpy(G["freq_x"]).cuda().requires_grad_(True)
    yf = FreqEncoder(input_dim=3, degree=4)(x)
    _close(yf, G["freq_y"], rtol=1e-5, atol=2e-6)
    yf.backward(_seeded(yf.shape, 24, -1, 1).cuda())
    _close(x.grad, G["freq_gx"], rtol=1e-4, atol=1e-4)
    net = FFMLP(32, 3, 64, 3).cuda()
    assert np.array_equal(net.weights.detach().cpu().numpy(), G["ffmlp_w"])  # seed-42 init, same RNG consumption
    net.train()
    yy = net(torch.from_numpy(G["ffmlp_x"]).cuda())
    _close(yy, G["ffmlp_y"], rtol=2e-2, atol=2e-2)  # fp16 MLP: fp32 accumulation here, fp16 rounding of activations in both


def test_march_wrapper_vs_reference_fixture(hip, G, cpu_random):
    import raymarching.raymarching as rm
    from nerf import synthetic as syn
    _, bits = syn.lego_like_density_grid(seed=0)
    ro, rd = torch.from_numpy(G["march_ro"]).cuda(), torch.from_numpy(G["march_rd"]).cuda()
    nears, fars = rm.near_far_from_aabb(ro, rd, torch.tensor([-1.0, -1, -1, 1, 1, 1]).cuda(), 0.2)
    assert np.array_equal(nears.cpu().numpy(), G["march_nears"]) and np.array_equal(fars.cpu().numpy(), G["march_fars"])
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    torch.manual_seed(5)
    xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, 1.0, torch.from_numpy(bits).cuda(), 1, 128, nears, fars, counter, -1, True,
                                                   128, False, 0, 1024)
    assert np.array_equal(counter.cpu().numpy(), G["march_counter"]) and np.array_equal(rays.cpu().numpy(), G["march_rays"])
    assert list(xyzs.shape) == G["march_xyzs_shape"].tolist()
    assert np.array_equal(xyzs[:256].cpu().numpy(), G["march_xyzs_head"]) and np.array_equal(deltas[:256].cpu().numpy(), G["march_deltas_head"])
    assert np.array_equal(xyzs.double().sum(0).cpu().numpy(), G["march_xyzs_sum"])
    assert np.array_equal(deltas.double().sum(0).cpu().numpy(), G["march_deltas_sum"])


def test_renderer_vs_reference_fixture(hip, G, cpu_random):
    """update_extra_state (full sweep, then partial), run_cuda training branch and inference loop (device compaction)"""
    from nerf import renderer, synthetic as syn
    lo, hi = syn.lego_like_boxes(0)

    class Analytic(renderer.NeRFRenderer):
        def forward(self, x, dd):
            return syn.box_density(x, lo, hi, sigma=40.0), (x * 0.5 + 0.5).clamp(0, 1) * (0.5 + 0.5 * dd.abs())

        def density(self, x):
            return {"sigma": syn.box_density(x, lo, hi, sigma=40.0)}
    ro, rd = torch.from_numpy(G["march_ro"]).cuda(), torch.from_numpy(G["march_rd"]).cuda()
    R = Analytic(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10).cuda()
    R.train()
    torch.manual_seed(7)
    R.update_extra_state()
    tr = R.run_cuda(ro[None], rd[None], bg_color=1, perturb=True, max_steps=1024)
    torch.manual_seed(8)
    R.update_extra_state()
    R.eval()
    ev = R.run_cuda(ro[None], rd[None], bg_color=1, perturb=False, max_steps=1024)
    assert np.array_equal(R.density_bitfield.cpu().numpy(), G["rend_bitfield"])
    assert R.mean_count == int(G["rend_mean_count"]) and R.iter_density == int(G["rend_iter_density"])
    assert abs(R.mean_density - float(G["rend_mean_density"])) <= 1e-6 * abs(float(G["rend_mean_density"]))  # reduction order
    assert np.array_equal(R.step_counter.cpu().numpy(), G["rend_step_counter"])
    for got, key in ((tr["image"][0], "rend_train_image"), (tr["depth"][0], "rend_train_depth"), (ev["image"][0], "rend_eval_image"),
                     (ev["depth"][0], "rend_eval_depth")):
        assert _relmax(got, G[key]) < 1e-4, key
        _close(got, G[key], rtol=1e-4, atol=1e-5, what=key)


def test_sampling_path_without_occupancy_grid_vs_reference_fixture(hip, G, cpu_random):
    """`NeRFRenderer.run` (cuda_ray off: near/far by the HIP kernel, stratified + importance sampling and compositing as torch
    ops on the GPU; nerf/renderer.py:136-253) against the reference's own `run` executed on the CPU oracle (`run_*` arrays,
    the same fixture tests/test_golden_wrappers.py holds bit-exactly on CPU): BASELINE configs[0]'s path on the HIP side.
    1e-4 of the value range for image / depth / weights; a sample that sits within an ulp of a pdf bin edge may be drawn
    from the neighbouring bin under the GPU's cumsum order, so up to 0.2 % of the pixels may differ by more (<= 2e-2)."""
    from nerf import renderer, synthetic as syn
    lo, hi = syn.lego_like_boxes(0)

    class AnalyticRun(renderer.NeRFRenderer):
        def density(self, x):
            return {"sigma": syn.box_density(x, lo, hi, sigma=40.0)}

        def color(self, x, dd, mask=None, **kw):
            rgb = (x * 0.5 + 0.5).clamp(0, 1) * (0.5 + 0.5 * dd.abs())
            if mask is None:
                return rgb
            out = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            out[mask] = rgb[mask]
            return out
    ro, rd = torch.from_numpy(G["march_ro"]).cuda(), torch.from_numpy(G["march_rd"]).cuda()
    R = AnalyticRun(bound=1, cuda_ray=False, density_scale=1, min_near=0.2).cuda()
    R.eval()
    ev = R.render(ro[None], rd[None], staged=True, max_ray_batch=1500, bg_color=1, perturb=False, num_steps=64, upsample_steps=48)
    R.train()
    torch.manual_seed(11)
    tr = R.render(ro[None, :1024], rd[None, :1024], bg_color=1, perturb=True, num_steps=64, upsample_steps=48)

    def check(got, key):
        got, ref = got.detach().float().cpu().numpy(), np.asarray(G[key])
        assert got.shape == ref.shape, key
        assert np.array_equal(np.isnan(got), np.isnan(ref)), key  # (rays that miss the box: 0 / 0 depth on both sides)
        d = np.abs(np.nan_to_num(got) - np.nan_to_num(ref))
        scale = max(float(np.nanmax(np.abs(ref))), 1e-30)
        assert float((d > 1e-4 * scale).mean()) <= 2e-3, (key, float((d > 1e-4 * scale).mean()), float(d.max()))
        assert float(d.max()) <= 2e-2 * scale, (key, float(d.max()))
    check(ev["image"][0], "run_eval_image")
    check(ev["depth"][0], "run_eval_depth")
    check(tr["image"][0], "run_train_image")
    check(tr["depth"][0], "run_train_depth")
    check(tr["weights_sum"], "run_train_weights_sum")
    assert float(torch.nan_to_num(ev["image"]).std()) > 0.05


def test_network_vs_reference_fixture(hip, G):
    from nerf import network
    net = network.NeRFNetwork(bound=1, cuda_ray=True, log2_hashmap_size=14).cuda()
    assert [k for k, _ in net.named_parameters()] == G["net_param_names"].tolist()
    for k, p in net.named_parameters():
        p.data.copy_(_seeded(p.shape, zlib.crc32(k.encode()) % 1000, -0.5, 0.5))
    sigma, color = net(torch.from_numpy(G["net_x"]).cuda(), torch.from_numpy(G["net_d"]).cuda())
    _close(sigma, G["net_sigma"], rtol=1e-4, atol=1e-6)
    _close(color, G["net_color"], rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ trainstep.npz
@pytest.fixture(scope="module")
def T():
    return np.load(os.path.join(GOLDEN, "trainstep.npz"))


NET = dict(bound=1, cuda_ray=True, log2_hashmap_size=14, density_scale=1, min_near=0.2, density_thresh=10)


def _golden_student():
    from nerf import network, synthetic as syn
    net = network.NeRFNetwork(**NET).cuda()
    for k, p in net.named_parameters():
        p.data.copy_(_seeded(p.shape, zlib.crc32(k.encode()) % 1000, -0.5, 0.5))
    dens, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(dens))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    return net


def _check_grads(net, T, prefix, frozen=()):
    worst = 0.0
    for k, p in net.named_parameters():
        key = f"{prefix}_{k.replace('.', '_')}"
        if k in frozen:
            assert key + "_none" in T.files and p.grad is None, k
            continue
        g = p.grad.detach().double().cpu()
        ref_norm = float(T[key + "_norm"])
        assert abs(float(g.norm()) - ref_norm) <= 1e-4 * ref_norm, (k, float(g.norm()), ref_norm)
        if key in T.files:
            ref = T[key]
            err = np.abs(g.numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
        else:
            ref = T[key + "_at_rows"]
            err = np.abs(g[torch.from_numpy(T[key + "_rows"])].numpy() - ref).max() / max(np.abs(T[key + "_at_rows"]).max(), 1e-30)
        worst = max(worst, err)
        assert err < 2e-4, (k, err)
    return worst


def test_finetune_step_loss_and_gradients_vs_reference(hip, T, cpu_random):
    """One Seal fine-tuning loss (MSE(rgb) + L1(depth), nerf/utils.py:436-537) of the two-encoder network on the HIP path,
    fp32: loss, predicted colours, sample count and every parameter gradient against the reference's own train_step."""
    from sealnerf import SealTrainer
    net = _golden_student()
    net.mean_count = int(T["ts_mean_count"])
    tr = SealTrainer(net, net, lr=1e-2, fp16=False, native_optim=False)
    net.train()
    ro, rd = torch.from_numpy(T["ts_rays_o"]).cuda(), torch.from_numpy(T["ts_rays_d"]).cuda()
    torch.manual_seed(5)
    loss, out = tr.finetune_loss(ro, rd, torch.from_numpy(T["ts_images"]).cuda(), torch.from_numpy(T["ts_depths"]).cuda(), bg_color=1)
    assert np.array_equal(net.step_counter[0].cpu().numpy(), T["ts_counter"]), "ray compaction / sample count"
    assert abs(float(loss) - float(T["ts_loss"])) <= 1e-5 * float(T["ts_loss"])
    assert _relmax(out["image"], T["ts_pred"]) < 1e-4
    net.zero_grad()
    loss.backward()
    _check_grads(net, T, "ts_grad")
    # plain NGP loss (no depth target)
    net.local_step = 0
    torch.manual_seed(5)
    loss_rgb, _ = tr.finetune_loss(ro, rd, torch.from_numpy(T["ts_images"]).cuda(), None, bg_color=1)
    assert abs(float(loss_rgb) - float(T["ts_loss_rgb_only"])) <= 1e-5 * float(T["ts_loss_rgb_only"])


def test_pretrain_step_loss_and_gradients_vs_reference(hip, T):
    """SealNeRF/trainer.py:455-488: L1(sigma) + L1(colour) with frozen MLPs — only the two hash tables receive gradients"""
    from sealnerf import SealTrainer
    net = _golden_student()
    tr = SealTrainer(net, net, lr=1e-2, fp16=False, native_optim=False)
    net.train()
    tr.freeze_mlp(True)
    frozen = [k for k, p in net.named_parameters() if not p.requires_grad]
    assert frozen == T["pt_frozen"].tolist()
    net.zero_grad()
    loss = tr.pretrain_loss(torch.from_numpy(T["pt_points"]).cuda(), torch.from_numpy(T["pt_dirs"]).cuda(),
                            torch.from_numpy(T["pt_sigma"]).cuda(), torch.from_numpy(T["pt_color"]).cuda())
    assert abs(float(loss) - float(T["pt_loss"])) <= 1e-5 * float(T["pt_loss"])
    loss.backward()
    _check_grads(net, T, "pt_grad", frozen=frozen)


def test_eval_render_and_psnr_vs_reference(hip, T):
    """64x64 inference render of the same weights: HIP path vs the reference renderer on the CPU oracle; PSNR with the
    reference's PSNRMeter formula (nerf/utils.py:226-233) within 0.1 dB (north_star), pixels within 1e-4 relative."""
    from nerf.trainer import psnr
    net = _golden_student()
    net.eval()
    with torch.no_grad():
        ev = net.render(torch.from_numpy(T["ev_rays_o"]).cuda(), torch.from_numpy(T["ev_rays_d"]).cuda(), bg_color=1, perturb=False,
                        max_steps=1024, T_thresh=1e-4, dt_gamma=0)
    truth = torch.from_numpy(T["ev_truth"]).cuda()
    p_hip = psnr(ev["image"], truth)
    assert abs(p_hip - float(T["ev_psnr"])) <= 0.1, (p_hip, float(T["ev_psnr"]))
    assert abs(p_hip - float(T["ev_psnr"])) <= 1e-3  # in fact far inside the bar
    assert psnr(ev["image"], torch.from_numpy(T["ev_image"]).cuda()) > 80.0  # HIP render vs reference-path render, dB
    assert _relmax(ev["image"], T["ev_image"]) < 1e-4 and _relmax(ev["depth"], T["ev_depth"]) < 1e-4


# ------------------------------------------------------------------------------------------------ seal_bbox.npz
@pytest.mark.parametrize("tag", ["both", "to", "from_rot"])
def test_seal_bbox_kernel_vs_reference_map_to_origin(hip, tag):
    """csrc/seal.hip (and the mapper's constants) vs SealNeRF/seal_utils.py map_to_origin executed on the same points"""
    from test_seal_golden import case_config
    from sealnerf import SealBBoxMapper
    S = np.load(os.path.join(GOLDEN, "seal_bbox.npz"))
    mapper = SealBBoxMapper(case_config(tag, S))
    pts, dirs = torch.from_numpy(S[f"{tag}_points"]).cuda(), torch.from_numpy(S[f"{tag}_dirs"]).cuda()
    mapper.native = True
    p, d, m = mapper.map_to_origin(pts, dirs)
    ref_m = torch.from_numpy(S[f"{tag}_mask"])
    # a point within rounding of a face may fall on either side: none does in this seeded set
    assert torch.equal(m.cpu(), ref_m)
    _close(p, S[f"{tag}_out_points"], rtol=1e-5, atol=1e-6)
    _close(d, S[f"{tag}_out_dirs"], rtol=1e-5, atol=1e-6)


def test_finetune_step_fp16_fused_path_vs_reference(hip, T, cpu_random):
    """The same fine-tuning loss on the `-O` path (fp16 autocast: fp16 tables, MFMA MLPs, glue kernels, level-major hand-over)
    against the reference's fp32 result: the integer outcome (marching, sample count) is identical, the loss and the image
    agree to fp16 accuracy — the bound is the precision of the reference's own `-O` mode, not of this build."""
    from sealnerf import SealTrainer
    net = _golden_student()
    net.mean_count = int(T["ts_mean_count"])
    tr = SealTrainer(net, net, lr=1e-2, fp16=True, native_optim=False)
    net.train()
    ro, rd = torch.from_numpy(T["ts_rays_o"]).cuda(), torch.from_numpy(T["ts_rays_d"]).cuda()
    torch.manual_seed(5)
    with torch.autocast("cuda", dtype=torch.float16):
        assert net._can_fuse(torch.zeros(128, 3, device="cuda"))
    loss, out = tr.finetune_loss(ro, rd, torch.from_numpy(T["ts_images"]).cuda(), torch.from_numpy(T["ts_depths"]).cuda(), bg_color=1)
    assert np.array_equal(net.step_counter[0].cpu().numpy(), T["ts_counter"])
    assert abs(float(loss.detach()) - float(T["ts_loss"])) <= 5e-3 * float(T["ts_loss"])
    assert _relmax(out["image"], T["ts_pred"]) < 2e-2
    net.zero_grad()
    loss.backward()
    for k, p in net.named_parameters():
        ref = float(T[f"ts_grad_{k.replace('.', '_')}_norm"])
        assert abs(float(p.grad.double().norm()) - ref) <= 3e-2 * ref, (k, float(p.grad.double().norm()), ref)


# ----------------------------------------------------------------------------- integer kernels vs the reference TEXT
# tests/golden/int_kernels.npz: raymarching.cu:42-81 and gridencoder.cu:50-84 (+ the index lines of kernel_grid) evaluated
# statement by statement by oracle/gen_golden.py `int` (CPU twin of these tests: tests/test_int_golden.py).
@pytest.fixture(scope="module")
def GI():
    return np.load(os.path.join(GOLDEN, "int_kernels.npz"))


def _hip_morton(hip, coords):
    c = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.int32)).cuda()
    out = torch.empty(c.shape[0], dtype=torch.int32, device="cuda")
    hip.RaymarchingBackend.morton3D(c, c.shape[0], out)
    return out.cpu().numpy()


def _hip_invert(hip, ind):
    i = torch.from_numpy(np.ascontiguousarray(ind, dtype=np.int32)).cuda()
    out = torch.empty(i.shape[0], 3, dtype=torch.int32, device="cuda")
    hip.RaymarchingBackend.morton3D_invert(i, i.shape[0], out)
    return out.cpu().numpy()


def test_morton_kernels_reproduce_the_reference_text(hip, GI):
    v = GI["expand_in"]
    c = np.zeros((v.size, 3), np.int32)
    c[:, 0] = v.view(np.int32)                      # `__morton3D(x, 0, 0) == __expand_bits(x)`, wrap-around above 10 bits included
    assert np.array_equal(_hip_morton(hip, c).view(np.uint32), GI["expand_out"])
    assert np.array_equal(_hip_morton(hip, GI["morton_coords"]), GI["morton_indices"])
    g = np.arange(128, dtype=np.int32)
    sweep = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    assert np.uint32(zlib.crc32(_hip_morton(hip, sweep).tobytes())) == GI["morton_sweep128_crc"]
    assert np.array_equal(_hip_invert(hip, GI["invert_indices"]), GI["invert_coords"])   # negative int32 (arithmetic shift) included
    assert np.uint32(zlib.crc32(_hip_invert(hip, np.arange(128 ** 3, dtype=np.int32)).tobytes())) == GI["invert_sweep128_crc"]


@pytest.mark.parametrize("C", [1, 2, 4, 8])
def test_cascade_selection_reproduces_the_reference_text(hip, GI, C):
    xyz, dt = GI["mip_xyz"], GI["mip_dt"]
    n = max(xyz.shape[0], dt.shape[0])
    xyz_p = np.zeros((n, 3), np.float32)
    xyz_p[:xyz.shape[0]] = xyz
    dt_p = np.ones(n, np.float32)
    dt_p[:dt.shape[0]] = dt
    mp, md = hip.RaymarchingBackend.mip_levels(torch.from_numpy(xyz_p).cuda(), torch.from_numpy(dt_p).cuda(), 128, C)
    assert np.array_equal(mp.cpu().numpy()[:xyz.shape[0]], GI[f"mip_pos_c{C}"])
    assert np.array_equal(md.cpu().numpy()[:dt.shape[0]], GI[f"mip_dt_c{C}"])


@pytest.mark.parametrize("tag", ["lego", "hash", "smooth", "tiled_ac"])
def test_grid_corner_rows_reproduce_the_reference_text(hip, GI, tag):
    """scale table (host), cell of a point and the table row of each of its 2^D corners, per level: `get_grid_index` /
    `fast_hash` (gridencoder.cu:50-84) applied to the cells `kernel_grid` (:137-149) derives — hashed, dense and tiled levels."""
    D, C, gridtype, ac, L, H = GI[f"grid_{tag}_cfg"].tolist()
    S = float(GI[f"grid_{tag}_S"])
    assert np.array_equal(np.asarray(hip.level_scales(L, S, H), dtype=np.float32), GI[f"grid_{tag}_scales"])
    x = torch.from_numpy(GI[f"grid_{tag}_x"]).cuda()
    offsets = torch.from_numpy(GI[f"grid_{tag}_offsets"]).cuda()
    B = x.shape[0]
    cidx = torch.empty(B, L, 1 << D, dtype=torch.int32, device="cuda")
    hip.GridBackend.grid_corner_indices(x, offsets, cidx, B, D, C, L, S, H, gridtype, bool(ac))
    rows = cidx.cpu().numpy().view(np.uint32).astype(np.int64)
    assert np.array_equal(rows * C, GI[f"grid_{tag}_index"].astype(np.int64))


def test_packbits_and_near_far_reproduce_the_reference_text(hip, GI):
    """kernel_packbits (raymarching.cu:262-289) and kernel_near_far_from_aabb (:92-145) evaluated from the reference text
    (oracle/gen_golden.py `int`): the HIP kernels bit for bit — threshold neighbours, +-0 / +-inf / NaN cells; hits, misses
    (FLT_MAX), axis-parallel rays, origins on a slab plane, both min_near values"""
    cells = GI["packbits_grid"]
    grid = torch.from_numpy(np.ascontiguousarray(cells.reshape(-1))).cuda()
    for th in (10.0, 0.0, 0.01):
        bf = torch.zeros(cells.shape[0], dtype=torch.uint8, device="cuda")
        hip.RaymarchingBackend.packbits(grid, cells.shape[0], th, bf)
        assert np.array_equal(bf.cpu().numpy(), GI[f"packbits_thresh{th:g}"]), th
    ro, rd = torch.from_numpy(GI["nearfar_rays_o"]).cuda(), torch.from_numpy(GI["nearfar_rays_d"]).cuda()
    aabb = torch.from_numpy(GI["nearfar_aabb"]).cuda()
    N = ro.shape[0]
    for mn in (0.2, 0.05):
        nears, fars = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        hip.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, N, mn, nears, fars)
        got = np.stack([nears.cpu().numpy(), fars.cpu().numpy()], -1)
        want = GI[f"nearfar_min{mn:g}"]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.argwhere(got.view(np.uint32) != want.view(np.uint32))[:5]


# ----------------------------------------------------------------------------- compositing vs the reference TEXT (1e-4)
# tests/golden/float_kernels.npz: kernel_composite_rays_train_forward / _backward and kernel_composite_rays run statement by
# statement in float32 (oracle/gen_golden.py `float`; CPU twin: tests/test_float_golden.py, whose helpers are used here).
def test_training_compositing_within_1e4_of_the_reference_text(hip):
    import test_float_golden as fg
    G = np.load(os.path.join(GOLDEN, "float_kernels.npz"))
    fg.check_train(fg.train_compositing(hip.RaymarchingBackend, G, dev="cuda"), G)


def test_inference_compositing_within_1e4_of_the_reference_text(hip):
    import test_float_golden as fg
    G = np.load(os.path.join(GOLDEN, "float_kernels.npz"))
    fg.check_inference(fg.inference_compositing(hip.RaymarchingBackend, G, dev="cuda"), G)


# ----------------------------------------------------------------------------- the marchers vs the reference TEXT, bit for bit
# tests/golden/march_kernels.npz: kernel_march_rays_train and kernel_march_rays run statement by statement with nvcc's
# multiply-add contraction modelled (oracle/gen_golden.py `march`; CPU twin and helpers: tests/test_march_golden.py).
@pytest.mark.parametrize("tag", ["c1", "c2", "c1_noperturb"])
@pytest.mark.parametrize("path", [1, 2, 3], ids=["lane-per-ray", "wave-per-ray", "wave-per-ray-general"])
def test_training_marcher_reproduces_the_reference_text(hip, tag, path):
    """ray table (ray-ordered spans), counter, every sample's position / direction / deltas — all three marching kernels (lane
    per ray, wave per ray with the single-cascade voxel-run fast path, wave per ray general); one cascade and two, dt_gamma 0
    and 1/128, perturbed and unperturbed starts, a sample buffer the last rays do not fit"""
    import test_march_golden as mg
    G = np.load(os.path.join(GOLDEN, "march_kernels.npz"))
    R = hip.RaymarchingBackend
    old = R._march_path
    R._march_path = path
    try:
        mg.check_march(mg.run_march(R, G, tag, dev="cuda"), G, tag)
    finally:
        R._march_path = old


def test_inference_marcher_reproduces_the_reference_text(hip):
    import test_march_golden as mg
    G = np.load(os.path.join(GOLDEN, "march_kernels.npz"))
    mg.check_march_infer(mg.run_march_infer(hip.RaymarchingBackend, G, dev="cuda"), G)


@pytest.mark.parametrize("tag", ["hash", "smooth", "tiled_ac", "lego"])
def test_grid_forward_reproduces_the_reference_text(hip, tag):
    """tests/golden/grid_kernels.npz: `kernel_grid` (gridencoder.cu:87-242) run statement by statement with nvcc's contraction
    modelled (oracle/gen_golden.py `grid`; CPU twin: tests/test_grid_golden.py): fp32 outputs bit for bit, dy_dx to 1e-6"""
    import test_grid_golden as gg
    G = np.load(os.path.join(GOLDEN, "grid_kernels.npz"))
    gg.check_forward(gg.run_forward(hip.GridBackend, G, tag, dev="cuda"), G, tag)


@pytest.mark.parametrize("tag", ["hash", "smooth", "tiled_ac"])
def test_grid_backward_within_summation_order_of_the_reference_text(hip, tag):
    """kernel_grid_backward / kernel_input_backward (gridencoder.cu:245-366) run statement by statement: the same rows touched,
    sums within fp32 summation order (2e-6 of the largest entry), input gradient from the forward's dy_dx"""
    import test_grid_golden as gg
    G = np.load(os.path.join(GOLDEN, "grid_kernels.npz"))
    gg.check_backward(gg.run_backward(hip.GridBackend, G, tag, dev="cuda"), G, tag)


@pytest.mark.parametrize("tag", ["hash", "lego"])
def test_grid_forward_fp16_tables_reproduce_the_reference_text(hip, tag):
    """the `-O` instantiation of kernel_grid (scalar_t = at::Half, c10::Half's operator semantics modelled): fp16 outputs of the HIP
    forward bit for bit, on the small hash configuration and on the Lego table"""
    import test_grid_golden as gg
    G = np.load(os.path.join(GOLDEN, "grid_kernels.npz"))
    gg.check_forward_f16(gg.run_forward_f16(hip.GridBackend, G, tag, dev="cuda"), G, tag)


# ---- the remaining native kernels against the reference TEXT: frequency encoder, sph_from_ray, grad_total_variation
# (oracle/gen_golden.py `enc`; CPU twin and tolerances: tests/test_enc_golden.py, whose helpers are used here)
@pytest.mark.parametrize("tag", ["tensorf", "dirs"])
def test_hip_freq_encoder_vs_reference_text(hip, tag):
    import test_enc_golden as eg
    G = np.load(os.path.join(GOLDEN, "encoder_kernels.npz"))
    eg.check_freq(eg.run_freq(hip.FreqBackend, G, tag, dev="cuda"), G, tag)


def test_hip_sph_from_ray_vs_reference_text(hip):
    import test_enc_golden as eg
    G = np.load(os.path.join(GOLDEN, "encoder_kernels.npz"))
    eg.check_sph(eg.run_sph(hip.RaymarchingBackend, G, dev="cuda"), G)


@pytest.mark.parametrize("tag", ["hash", "tiled_ac"])
def test_hip_grad_total_variation_vs_reference_text(hip, tag):
    import test_enc_golden as eg
    G = np.load(os.path.join(GOLDEN, "encoder_kernels.npz"))
    eg.check_tv(eg.run_tv(hip.GridBackend, G, tag, dev="cuda"), G, tag)


@pytest.mark.parametrize("path", [1, 2], ids=["atomics", "binned"])
def test_hip_grid_backward_fp16_contributions_of_the_reference_text(hip, path):
    """the `-O` branch of kernel_grid_backward (`(__half)(w * grad)` per contribution, `__half2` atomics): the binned path sums the
    contributions exactly (64-bit fixed point) and must reproduce the correctly rounded exact sum of the reference's half values
    (consecutive points of one cell are merged in fp32 before the rounding: at most an ulp on the few rows where that happens);
    the atomics path adds in fp16 in the hardware's order and stays within the drift thread order shows"""
    import test_grid_golden as gg
    G = np.load(os.path.join(GOLDEN, "grid_kernels.npz"))
    hip.GridBackend.set_backward_path(path)
    try:
        got = gg.run_backward_f16(hip.GridBackend, G, "hash", dev="cuda")
    finally:
        hip.GridBackend.set_backward_path(0)
    if path == 2:
        exact, rounded = G["hash_grad_emb_f16_exact"], G["hash_grad_emb_f16_exact_rounded"]
        assert np.array_equal(got == 0, rounded == 0)
        bad = got.view(np.uint16) != rounded.view(np.uint16)
        one_ulp = np.abs(got.astype(np.float64) - exact) <= np.abs(np.spacing(rounded)).astype(np.float64)
        assert one_ulp.all() and bad.mean() < 0.02, (int(bad.sum()), got.size)
    else:
        gg.check_backward_f16(got, G, "hash", exact_sum=False)
