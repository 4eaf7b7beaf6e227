"""GPU twin of tests/test_seal_loop_golden.py: the build's Seal caller side on libseal3d_hip.so against the outputs of the
REFERENCE's SealNeRF/renderer.py / trainer.py / provider.py executed on the CPU oracle (tests/golden/seal_loop.npz,
oracle/gen_golden.py `seal_loop`).  Bars (north_star): integers — forced cells, bitfield, sample counters, the alive-ray
compaction trace — exact; composited colour / depth and the distillation targets within 1e-4 relative."""
import numpy as np
import pytest
import torch

from test_seal_loop_golden import CASES, OPT, G, _distillation, case_mapper, golden_network, relmax  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", CASES)
def test_teacher_render_through_the_proxy_vs_reference(hip, G, tag):
    """SealNeRF/renderer.py:22-66, 254-418 on the HIP path (native bbox mapper, fp32)"""
    from sealnerf import make_teacher
    mapper = case_mapper(tag)
    teacher = golden_network(make_teacher, mapper, "cuda")
    assert np.array_equal(mapper.map_data["force_fill_bound"].cpu().numpy(), G[f"{tag}_fill_bound_clamped"])
    assert np.array_equal(teacher.force_fill_grid_indices.cpu().numpy(), G[f"{tag}_grid_indices"])
    assert np.array_equal(teacher.force_fill_bitfield_indices.cpu().numpy(), G[f"{tag}_bitfield_indices"])
    teacher.hack_bitfield()
    assert np.array_equal(teacher.density_bitfield.cpu().numpy(), G[f"{tag}_bitfield_hacked"])
    ro, rd = torch.from_numpy(G["rays_o"]).cuda(), torch.from_numpy(G["rays_d"]).cuda()
    teacher.train()
    with torch.no_grad():
        tr = teacher.render(ro, rd, staged=True, bg_color=None, perturb=False, force_all_rays=True, **OPT)
    assert np.array_equal(teacher.step_counter[0].cpu().numpy(), G[f"{tag}_train_counter"]), "sample count of the un-budgeted march"
    for k in ("image", "depth", "weights_sum"):
        assert relmax(tr[k].reshape(G[f"{tag}_train_{k}"].shape).cpu(), G[f"{tag}_train_{k}"]) < 1e-4, k
    teacher.eval()
    # the reference's loop shape (host compaction by boolean mask, 8 steps per iteration at most): the trace is comparable
    teacher.device_compaction = False
    import nerf.renderer as rend
    rm = rend.raymarching
    trace, real = [], rm.march_rays
    rm.march_rays = lambda n_alive, n_step, *a, **k: (trace.append((n_alive, n_step)), real(n_alive, n_step, *a, **k))[1]
    try:
        with torch.no_grad():
            ev = teacher.render(ro, rd, staged=True, bg_color=None, perturb=False, force_all_rays=True, **OPT)
    finally:
        rm.march_rays = real
    assert np.array_equal(np.array(trace), G[f"{tag}_eval_trace"]), "alive-ray compaction trace"
    assert relmax(ev["image"].cpu(), G[f"{tag}_eval_image"]) < 1e-4 and relmax(ev["depth"].cpu(), G[f"{tag}_eval_depth"]) < 1e-4
    # the product's default loop (device-side compaction, count kept on the device): the same frame
    teacher.device_compaction = True
    with torch.no_grad():
        ev2 = teacher.render(ro, rd, staged=True, bg_color=None, perturb=False, force_all_rays=True, **OPT)
    assert relmax(ev2["image"].cpu(), G[f"{tag}_eval_image"]) < 1e-4 and relmax(ev2["depth"].cpu(), G[f"{tag}_eval_depth"]) < 1e-4


@pytest.mark.parametrize("native_optim", [False, True])
def test_init_pretraining_and_two_epochs_vs_reference(hip, G, native_optim):
    """SealNeRF/trainer.py:88-263, 363-503: lattices exact, teacher targets 1e-4, per-step losses of two epochs of frozen-MLP
    Adam and the student's tables afterwards (torch.optim.Adam and the HIP multi-tensor Adam)"""
    teacher, student, tr, mapper = _distillation(G, "cuda", native_optim=native_optim)
    torch.manual_seed(11)  # (seed=None: the directions are drawn on the CPU generator, one randint per part, like the reference's)
    n = tr.init_pretraining(epochs=2, batch_size=3000, lr=0.02, local_point_step=0.02, local_angle_step=45,
                            surrounding_point_step=0.04, surrounding_angle_step=45, surrounding_bounds_extend=0.1,
                            global_point_step=0.25, global_angle_step=90, seed=None)
    assert list(tr.pretraining_data) == G["ip_parts"].tolist() and n == G["ip_local_points"].shape[0]
    assert np.array_equal(mapper.map_data["force_fill_bound"].cpu().numpy(), G["ip_fill_bound_after"])
    for part, src in tr.pretraining_data.items():
        assert src["steps"] == G[f"ip_{part}_steps"].tolist(), part
        assert np.array_equal(src["points"].cpu().numpy(), G[f"ip_{part}_points"]), part  # lattice + the mapper's inside test
        np.testing.assert_allclose(src["dirs"].cpu().numpy(), G[f"ip_{part}_dirs"], atol=1e-7, err_msg=part)
        assert relmax(src["sigma"].cpu(), G[f"ip_{part}_sigma"]) < 1e-4 and relmax(src["color"].cpu(), G[f"ip_{part}_color"]) < 1e-4, part
    losses = []
    for _ in range(2):
        tr.pretrain_one_epoch()
        losses += [float(l) for l in tr.last_pretrain_losses]
    np.testing.assert_allclose(losses, G["pe_losses"], rtol=1e-3)
    tr.end_pretraining()
    assert tr.optimizer.param_groups[0]["lr"] == float(G["pe_lr_after"])
    for k, p in student.named_parameters():
        key = f"pe_param_{k.replace('.', '_')}"
        v = p.detach().cpu()
        assert abs(float(v.double().norm()) - float(G[key + "_norm"])) <= 1e-4 * float(G[key + "_norm"]), k
        if key in G.files:
            assert relmax(v, G[key]) < 1e-4, k  # (frozen MLPs: untouched)
        else:
            # Adam's first steps move an entry by ~lr * sign(g): an entry whose gradient is rounding noise may land a step apart
            d = np.abs(v[torch.from_numpy(G[key + "_rows"])].numpy() - G[key + "_at_rows"])
            assert np.mean(d > 1e-3) < 0.01, (k, float(np.mean(d > 1e-3)))


def test_proxy_truth_and_provider_vs_reference(hip, G):
    """SealNeRF/trainer.py:506-586 + SealNeRF/provider.py:19-128 on the HIP path"""
    from nerf import synthetic as syn
    from sealnerf import SealDataset
    teacher, student, tr, mapper = _distillation(G, "cuda", native_optim=True)
    ro, rd = torch.from_numpy(G["rays_o"]).cuda(), torch.from_numpy(G["rays_d"]).cuda()
    data = {"rays_o": ro, "rays_d": rd, "images": torch.zeros(1, ro.shape[1], 3, device="cuda")}
    tr.proxy_truth_data(data)
    assert teacher.density_bitfield_hacked and not teacher.training
    assert relmax(data["images"].cpu(), G["pt_eval_images"]) < 1e-4 and relmax(data["depths"].cpu(), G["pt_eval_depths"]) < 1e-4
    data = {"rays_o": ro, "rays_d": rd, "images": torch.zeros(1, ro.shape[1], 3, device="cuda")}
    tr.proxy_truth_data(data, n_batch=5)
    assert relmax(data["images"].cpu(), G["pt_eval_images_nb5"]) < 1e-4 and relmax(data["depths"].cpu(), G["pt_eval_depths_nb5"]) < 1e-4
    teacher.train()
    data = {"rays_o": ro, "rays_d": rd, "images": torch.zeros(1, ro.shape[1], 3, device="cuda")}
    tr.proxy_truth_data(data)
    teacher.eval()
    assert relmax(data["images"].cpu(), G["pt_train_images"]) < 1e-4 and relmax(data["depths"].cpu(), G["pt_train_depths"]) < 1e-4
    tr.init_proxy_cache(2, 256)
    for name in ("a", "b"):
        pix = torch.from_numpy(G[f"pc_{name}_pixels"]).cuda()
        d_ = {"rays_o": torch.from_numpy(G["pc_rays_o"]).cuda()[:, pix[0]].contiguous(), "rays_d": torch.from_numpy(G["pc_rays_d"]).cuda()[:, pix[0]].contiguous(),
              "images": torch.zeros(1, pix.shape[1], 3, device="cuda"), "data_index": torch.tensor([1]), "pixel_index": pix}
        tr.proxy_truth_data(d_, use_cache=True)
        assert relmax(d_["images"].cpu(), G[f"pc_{name}_images"]) < 1e-4 and relmax(d_["depths"].cpu(), G[f"pc_{name}_depths"]) < 1e-4
    assert np.array_equal(tr.proxy_cache_mask.cpu().numpy(), G["pc_mask"])
    assert relmax(tr.proxy_cache_image.cpu(), G["pc_image"]) < 1e-4 and relmax(tr.proxy_cache_depth.cpu(), G["pc_depth"]) < 1e-4
    ds = SealDataset(torch.from_numpy(G["pd_poses"]).cuda(), syn.lego_intrinsics(24, 24), 24, 24, num_rays=96, render_kwargs=OPT)
    ds.proxy_dataset(teacher, n_batch=1)
    assert relmax(ds.images.cpu(), G["pd_images"]) < 1e-4 and relmax(ds.depths.cpu(), G["pd_depths"]) < 1e-4
    g = torch.Generator().manual_seed(21)
    batch = ds.collate([1], generator=g)
    assert np.array_equal(batch["pixel_index"].cpu().numpy(), G["pd_collate_inds"])
    for k in ("images", "depths", "rays_o", "rays_d"):
        assert relmax(batch[k].cpu(), G[f"pd_collate_{k}"]) < 1e-4, k


@pytest.mark.parametrize("name", ["hsv", "rgb", "both"])
def test_map_color_kernel_vs_reference_map_color(hip, name):
    """csrc/seal.hip s3d_seal_map_color vs the reference's `map_color` EXECUTED on seeded colours (tests/golden/seal_bbox.npz:
    greys, pure channels, channel ties, hue wrap-around) — all rows moved; then a partial mask against the torch route
    (`colors[mask] = map_color(colors[mask])`: the batch mean of the `rgb` edit is the moved rows' alone), fp32 and fp16"""
    import os
    from conftest import GOLDEN
    from sealnerf import SealBBoxMapper
    from test_seal import BBOX
    S = np.load(os.path.join(GOLDEN, "seal_bbox.npz"))
    opts = S[f"color_{name}_opts"]
    cfg = dict(BBOX)
    if not np.isnan(opts[0]).any():
        cfg["hsv"] = opts[0].tolist()
    if not np.isnan(opts[1]).any():
        cfg["rgb"], cfg["rgbLightOffset"] = opts[1].tolist(), float(opts[2][0])
    mapper = SealBBoxMapper(cfg)
    cols = torch.from_numpy(S["color_in"]).cuda()
    every = torch.ones(cols.shape[0], dtype=torch.bool, device="cuda")
    mapper.native = True
    out = mapper.map_color_masked(None, None, cols, every)
    ref = S[f"color_{name}"]
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6, float(np.abs(out.cpu().numpy() - ref).max())
    g = torch.Generator().manual_seed(3)
    part = (torch.rand(cols.shape[0], generator=g) < 0.4).cuda()
    a = mapper.map_color_masked(None, None, cols, part)
    mapper.native = False
    b = mapper.map_color_masked(None, None, cols, part)
    mapper.native = True
    assert torch.equal(a[~part], cols[~part]) and float((a - b).abs().max()) < 2e-6
    h = mapper.map_color_masked(None, None, cols.half(), part)
    assert h.dtype == torch.float16 and float((h.float() - b).abs().max()) < 4e-3
