"""The build's Seal caller side (seal-3d_amd/sealnerf/{renderer,trainer,provider}.py) on the CPU oracle against
tests/golden/seal_loop.npz — outputs of the REFERENCE's SealNeRF/renderer.py, SealNeRF/trainer.py and SealNeRF/provider.py
executed on the same oracle by oracle/gen_golden.py `seal_loop` (authoring container only).  Both sides run the same native
arithmetic here, so integers are exact and floats agree to summation-order noise; the GPU twin (HIP path, 1e-4) is
tests/test_gpu_seal_loop.py."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN

NET = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, log2_hashmap_size=14)
OPT = dict(dt_gamma=0, max_steps=1024, T_thresh=1e-4)
CASES = ["both", "to", "from_rot", "both_color"]


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "seal_loop.npz"))


def _seeded(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def case_mapper(tag):
    """the edit of a fixture case (oracle/gen_golden.py: SEAL_CASES / SEAL_LOOP_COLOR) as a fresh mapper"""
    from test_seal_golden import case_config
    from sealnerf import SealBBoxMapper
    S = np.load(os.path.join(GOLDEN, "seal_bbox.npz"))
    cfg = case_config(tag.replace("_color", ""), S)
    if tag.endswith("_color"):
        cfg.update(hsv=[0.12, -0.05, 0.03], rgb=[0.8, 0.2, 0.1], rgbLightOffset=0.05)
    return SealBBoxMapper(cfg)


def golden_network(make, mapper, device="cpu"):
    """the fixture's network: parameters seeded by name, lego-like occupancy, mapper attached"""
    from nerf import network, synthetic as syn
    net = make(network.NeRFNetwork, **NET).to(device)
    for k, p in net.named_parameters():
        p.data.copy_(_seeded(p.shape, zlib.crc32(k.encode()) % 1000, -0.5, 0.5))
    dens, bits = syn.lego_like_density_grid(seed=0)
    net.density_grid.copy_(torch.from_numpy(dens))
    net.density_bitfield.copy_(torch.from_numpy(bits))
    net.init_mapper(mapper)
    return net


def relmax(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("tag", CASES)
def test_init_mapper_hack_bitfield_and_teacher_render(oracle_wrappers, G, tag):
    """SealNeRF/renderer.py:22-66, 254-418: forced cells, the hacked bitfield, and both branches of the teacher's run_cuda"""
    from sealnerf import make_teacher
    mapper = case_mapper(tag)
    assert np.array_equal(mapper.map_data["force_fill_bound"].numpy(), G[f"{tag}_fill_bound_in"])
    teacher = golden_network(make_teacher, mapper)
    assert np.array_equal(mapper.map_data["force_fill_bound"].numpy(), G[f"{tag}_fill_bound_clamped"]), "in-place clamp (:31-32)"
    assert np.array_equal(teacher.force_fill_grid_indices.numpy(), G[f"{tag}_grid_indices"])
    assert np.array_equal(teacher.force_fill_bitfield_indices.numpy(), G[f"{tag}_bitfield_indices"])
    before = teacher.density_bitfield.clone()
    teacher.hack_bitfield()
    assert np.array_equal(teacher.density_bitfield.numpy(), G[f"{tag}_bitfield_hacked"])
    teacher.restore_bitfield()
    assert torch.equal(teacher.density_bitfield, before) and not teacher.density_bitfield_hacked
    teacher.hack_bitfield()
    ro, rd = torch.from_numpy(G["rays_o"]), torch.from_numpy(G["rays_d"])
    teacher.train()
    with torch.no_grad():
        tr = teacher.render(ro, rd, staged=True, bg_color=None, perturb=False, force_all_rays=True, **OPT)
    assert np.array_equal(teacher.step_counter[0].numpy(), G[f"{tag}_train_counter"])
    for k in ("image", "depth", "weights_sum"):
        assert relmax(tr[k].reshape(G[f"{tag}_train_{k}"].shape), G[f"{tag}_train_{k}"]) < 1e-6, k
    teacher.eval()
    import nerf.renderer as rend
    rm = rend.raymarching  # (the package object the renderer calls through)
    trace, real = [], rm.march_rays
    rm.march_rays = lambda n_alive, n_step, *a, **k: (trace.append((n_alive, n_step)), real(n_alive, n_step, *a, **k))[1]
    try:
        with torch.no_grad():
            ev = teacher.render(ro, rd, staged=True, bg_color=None, perturb=False, force_all_rays=True, **OPT)
    finally:
        rm.march_rays = real
    assert np.array_equal(np.array(trace), G[f"{tag}_eval_trace"]), "alive-ray compaction trace of the inference loop"
    assert relmax(ev["image"], G[f"{tag}_eval_image"]) < 1e-6 and relmax(ev["depth"], G[f"{tag}_eval_depth"]) < 1e-6


def _distillation(G, device="cpu", **trainer_kw):
    from sealnerf import SealTrainer, make_student, make_teacher
    mapper = case_mapper("both_color")
    teacher = golden_network(make_teacher, mapper, device)
    teacher.eval()  # main_SealNeRF.py:210
    student = golden_network(make_student, teacher.seal_mapper, device)
    tr = SealTrainer(student, teacher, lr=1e-2, fp16=False, **trainer_kw)
    tr.render_kwargs.update(OPT)
    return teacher, student, tr, mapper


def test_sample_points_vs_reference(G):
    from sealnerf import sample_points
    pts, dirs = sample_points(case_mapper("both_color").map_data["force_fill_bound"], 0.05, 90)
    # (the fixture's bounds were clamped by init_mapper first; this edit lies inside the box, so the clamp is the identity)
    assert pts.shape == G["sp_points"].shape and np.array_equal(pts.numpy(), G["sp_points"])
    np.testing.assert_allclose(dirs.numpy(), G["sp_dirs"], atol=1e-12)


def test_init_pretraining_and_two_epochs_vs_reference(oracle_wrappers, G):
    """SealNeRF/trainer.py:88-263 (three parts, teacher targets, in-place growth of the fill bound) and :363-503 (two epochs of
    frozen-MLP Adam over local -> surrounding -> global chunks): every step's loss and the student's parameters afterwards"""
    teacher, student, tr, mapper = _distillation(G, native_optim=False)
    torch.manual_seed(11)
    n = tr.init_pretraining(epochs=2, batch_size=3000, lr=0.02, local_point_step=0.02, local_angle_step=45,
                            surrounding_point_step=0.04, surrounding_angle_step=45, surrounding_bounds_extend=0.1,
                            global_point_step=0.25, global_angle_step=90, seed=None)
    assert list(tr.pretraining_data) == G["ip_parts"].tolist() and n == G["ip_local_points"].shape[0]
    assert np.array_equal(mapper.map_data["force_fill_bound"].numpy(), G["ip_fill_bound_after"])
    for part, src in tr.pretraining_data.items():
        assert src["steps"] == G[f"ip_{part}_steps"].tolist(), part
        assert np.array_equal(src["points"].numpy(), G[f"ip_{part}_points"]), part
        np.testing.assert_allclose(src["dirs"].numpy(), G[f"ip_{part}_dirs"], atol=1e-7, err_msg=part)
        assert relmax(src["sigma"], G[f"ip_{part}_sigma"]) < 1e-6 and relmax(src["color"], G[f"ip_{part}_color"]) < 1e-6, part
    losses = []
    for _ in range(2):
        tr.pretrain_one_epoch()
        losses += [float(l) for l in tr.last_pretrain_losses]
    np.testing.assert_allclose(losses, G["pe_losses"], rtol=2e-5)
    assert not any(p.requires_grad for p in student.sigma_net.parameters()), "an epoch leaves the MLPs frozen (reference: train() unfreezes)"
    tr.end_pretraining()
    assert all(p.requires_grad for p in student.parameters())
    assert tr.optimizer.param_groups[0]["lr"] == float(G["pe_lr_after"]), "set_lr(-1) after two epochs (the reference's cache quirk)"
    assert bool(student.density_bitfield_hacked) == bool(G["pe_bitfield_hacked"])
    for k, p in student.named_parameters():
        key = f"pe_param_{k.replace('.', '_')}"
        v = p.detach()
        assert abs(float(v.double().norm()) - float(G[key + "_norm"])) <= 1e-5 * float(G[key + "_norm"]), k
        if key in G.files:
            assert relmax(v, G[key]) < 1e-4, k
        else:
            assert relmax(v[torch.from_numpy(G[key + "_rows"])], G[key + "_at_rows"]) < 1e-3, k


def test_proxy_truth_vs_reference(oracle_wrappers, G):
    """SealNeRF/trainer.py:506-586: teacher mode honoured (eval = the inference loop), n_batch pieces, skip_proxy, a full
    frame's shape, the pixel cache"""
    teacher, student, tr, mapper = _distillation(G, native_optim=False)
    ro, rd = torch.from_numpy(G["rays_o"]), torch.from_numpy(G["rays_d"])
    assert not teacher.density_bitfield_hacked
    data = {"rays_o": ro, "rays_d": rd, "images": torch.zeros(1, ro.shape[1], 3)}
    tr.proxy_truth_data(data)
    assert teacher.density_bitfield_hacked and not teacher.training
    assert relmax(data["images"], G["pt_eval_images"]) < 1e-6 and relmax(data["depths"], G["pt_eval_depths"]) < 1e-6
    data = {"rays_o": ro, "rays_d": rd, "images": torch.zeros(1, ro.shape[1], 3)}
    tr.proxy_truth_data(data, n_batch=5)
    assert relmax(data["images"], G["pt_eval_images_nb5"]) < 1e-6 and relmax(data["depths"], G["pt_eval_depths_nb5"]) < 1e-6
    skipped = {"rays_o": ro, "rays_d": rd, "images": torch.full((1, ro.shape[1], 3), 0.25), "skip_proxy": True}
    tr.proxy_truth_data(skipped)
    assert "depths" not in skipped and float(skipped["images"].mean()) == 0.25
    full = {"rays_o": ro[:, :64], "rays_d": rd[:, :64], "images_shape": [1, 8, 8, 3]}
    tr.proxy_truth_data(full)
    assert full["images"].shape == G["pt_full_images"].shape and full["depths"].shape == G["pt_full_depths"].shape
    assert relmax(full["images"], G["pt_full_images"]) < 1e-6 and relmax(full["depths"], G["pt_full_depths"]) < 1e-6
    teacher.train()
    data = {"rays_o": ro, "rays_d": rd, "images": torch.zeros(1, ro.shape[1], 3)}
    tr.proxy_truth_data(data)
    assert teacher.training
    teacher.eval()
    assert relmax(data["images"], G["pt_train_images"]) < 1e-6 and relmax(data["depths"], G["pt_train_depths"]) < 1e-6
    # pixel cache
    tr.init_proxy_cache(2, 256)
    import nerf.renderer as rend
    rm = rend.raymarching
    first, real = [], rm.march_rays
    rm.march_rays = lambda n_alive, *a, **k: (first.append(n_alive), real(n_alive, *a, **k))[1]
    try:
        for name in ("a", "b"):
            pix = torch.from_numpy(G[f"pc_{name}_pixels"])
            d_ = {"rays_o": torch.from_numpy(G["pc_rays_o"])[:, pix[0]].contiguous(), "rays_d": torch.from_numpy(G["pc_rays_d"])[:, pix[0]].contiguous(),
                  "images": torch.zeros(1, pix.shape[1], 3), "data_index": torch.tensor([1]), "pixel_index": pix}
            first.clear()
            tr.proxy_truth_data(d_, use_cache=True)
            assert first[0] == int(G[f"pc_{name}_first_alive"]), "only rays without a cache entry are rendered"
            assert relmax(d_["images"], G[f"pc_{name}_images"]) < 1e-6 and relmax(d_["depths"], G[f"pc_{name}_depths"]) < 1e-6
    finally:
        rm.march_rays = real
    assert np.array_equal(tr.proxy_cache_mask.numpy(), G["pc_mask"])
    assert relmax(tr.proxy_cache_image, G["pc_image"]) < 1e-6 and relmax(tr.proxy_cache_depth, G["pc_depth"]) < 1e-6


def test_proxy_dataset_and_collate_vs_reference(oracle_wrappers, G):
    """SealNeRF/provider.py:19-128: both poses rendered through the eval-mode teacher, targets gathered per batch"""
    from nerf import synthetic as syn
    from sealnerf import SealDataset
    teacher, student, tr, mapper = _distillation(G, native_optim=False)
    teacher.hack_bitfield()
    ds = SealDataset(torch.from_numpy(G["pd_poses"]), syn.lego_intrinsics(24, 24), 24, 24, num_rays=96, render_kwargs=OPT)
    ds.proxy_dataset(teacher, n_batch=1)
    assert ds.proxy_flag == bool(G["pd_flag"]) and ds.images.shape == G["pd_images"].shape and ds.depths.shape == G["pd_depths"].shape
    assert relmax(ds.images, G["pd_images"]) < 1e-6 and relmax(ds.depths, G["pd_depths"]) < 1e-6
    torch.manual_seed(21)
    batch = ds.collate([1])
    assert np.array_equal(batch["pixel_index"].numpy(), G["pd_collate_inds"]) and bool(batch["skip_proxy"]) == bool(G["pd_collate_skip"])
    for k in ("images", "depths", "rays_o", "rays_d"):
        assert batch[k].shape == G[f"pd_collate_{k}"].shape and relmax(batch[k], G[f"pd_collate_{k}"]) < 1e-6, k
