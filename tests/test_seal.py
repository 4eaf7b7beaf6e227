"""Seal-3D distillation layer (configs 3/4) on the CPU oracle: proxy mapper semantics, teacher render through the
proxy, local pretraining step, sharded pretraining == un-sharded pretraining (2 gloo ranks)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

BBOX = {"type": "bbox",
        "raw": [[x, y, z] for x in (-0.2, 0.2) for y in (0.0, 0.3) for z in (-0.2, 0.2)],
        "transform": [[1, 0, 0, 0.3], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], "scale": [1, 1, 1], "boundType": "both"}


def test_bbox_mapper_semantics():
    from sealnerf import SealBBoxMapper
    m = SealBBoxMapper(BBOX)
    fb = m.map_data["force_fill_bound"]
    torch.testing.assert_close(fb[0], torch.tensor([[0.1, 0.0, -0.2], [0.5, 0.3, 0.2]]))     # target box
    torch.testing.assert_close(fb[1], torch.tensor([[-0.2, 0.0, -0.2], [0.2, 0.3, 0.2]]))    # source box
    pts = torch.tensor([[0.4, 0.1, 0.05],      # inside target only -> pulled back by the inverse translation
                        [-0.1, 0.1, 0.05],     # inside source only -> also mapped (boundType both): leaves the box
                        [0.15, 0.15, 0.05],    # inside both
                        [0.8, 0.8, 0.8],       # outside
                        [0.4, 0.0, 0.05]])     # on the boundary plane y = 0: a zero coordinate fails `points.all(1)`
    dirs = torch.nn.functional.normalize(torch.randn(5, 3), dim=-1)
    out, od, mask = m.map_to_origin(pts, dirs)
    assert mask.tolist() == [True, True, True, False, False]
    torch.testing.assert_close(out[:3], pts[:3] - torch.tensor([0.3, 0, 0]))
    assert torch.equal(out[3:], pts[3:]) and torch.equal(od, dirs)   # pure translation: directions unchanged
    # scaling edit: target = source scaled x2 about its centre
    cfg = dict(BBOX, transform=np.eye(4).tolist(), scale=[2, 2, 2], boundType="to")
    m2 = SealBBoxMapper(cfg)
    c = torch.tensor([0.0, 0.15, 0.0])
    p = (c + torch.tensor([0.3, 0.2, -0.3]))[None]
    o2, _, k2 = m2.map_to_origin(p, None)
    assert bool(k2[0])
    torch.testing.assert_close(o2[0], c + torch.tensor([0.15, 0.1, -0.15]))


def test_sample_points_lattice_and_dirs():
    from sealnerf import sample_points
    b = torch.tensor([[0.0, 0.0, 0.0], [0.1, 0.05, 0.02]])
    pts, dirs = sample_points(b, 0.01, 45)
    n = [len(torch.arange(float(b[0][k]), float(b[1][k]), step=0.01)) for k in range(3)]  # torch.arange float semantics, as in the reference
    assert pts.shape == (n[0] * n[1] * n[2], 3) and dirs.shape == (512, 3)
    torch.testing.assert_close(dirs.norm(dim=-1), torch.full((512,), 1 - 1e-5, dtype=torch.float64))
    from scipy.spatial.transform import Rotation
    e = torch.stack(torch.meshgrid(*[torch.arange(0, 360, 45)] * 3, indexing="ij"), -1).reshape(-1, 3).numpy()
    ref = Rotation.from_euler("xyz", e, degrees=True).apply(np.array([1 - 1e-5, 0, 0]))
    np.testing.assert_allclose(dirs.numpy(), ref, atol=1e-12)


def _nets(oracle_wrappers):
    from nerf import network
    from sealnerf import SealBBoxMapper, make_student, make_teacher
    torch.manual_seed(0)
    kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, log2_hashmap_size=14)
    teacher = make_teacher(network.NeRFNetwork, **kw)
    student = make_student(network.NeRFNetwork, **kw)
    for p in teacher.parameters():
        p.data.uniform_(-0.3, 0.3)
    student.load_state_dict(teacher.state_dict())
    mapper = SealBBoxMapper(BBOX)
    teacher.init_mapper(mapper)
    student.init_mapper(mapper)
    return teacher, student, mapper


def test_teacher_proxy_and_pretrain_step(oracle_wrappers):
    from sealnerf import SealTrainer
    teacher, student, mapper = _nets(oracle_wrappers)
    # force-filled cells of both boxes are occupied after hack_bitfield, the rest of the (empty) grid is not
    teacher.hack_bitfield()
    assert int(teacher.density_bitfield.sum()) == 255 * int(torch.unique(teacher.force_fill_bitfield_indices).numel())
    # the teacher sees the source content at the target location: query equivalence through the proxy
    x = torch.tensor([[0.4, 0.1, 0.05]])
    d = torch.tensor([[0.0, 0.0, 1.0]])
    mx, md, mk = teacher.map_samples(x, d)
    s_t, c_t = teacher(mx, md)
    s_src, c_src = teacher(x - torch.tensor([0.3, 0, 0]), d)
    assert torch.equal(s_t, s_src) and torch.equal(c_t, c_src)
    # local pretraining: targets come from the teacher; a few steps reduce the L1 distillation loss
    tr = SealTrainer(student, teacher, lr=1e-2, fp16=False)
    n = tr.init_pretraining(batch_size=4096, lr=0.05, local_point_step=0.04)
    assert n > 500 and tr.pretraining_data["local"]["sigma"].shape == (n,)
    l0 = float(tr.pretrain_one_epoch())
    for _ in range(3):
        l1 = float(tr.pretrain_one_epoch())
    assert l1 < l0
    # MLPs were frozen during pretraining (an epoch leaves them frozen, SealNeRF/trainer.py:383) and are trainable again once
    # the phase is closed — train() does that before every fine-tuning epoch (:338-339), here the next train_step would
    assert not any(p.requires_grad for p in student.sigma_net.parameters())
    tr.end_pretraining()
    assert all(p.requires_grad for p in student.sigma_net.parameters()) and tr.optimizer.param_groups[0]["lr"] == 0.05
    w0 = teacher.sigma_net[0].weight
    assert torch.equal(student.sigma_net[0].weight, w0), "frozen MLP weights must not move during local pretraining"
    assert not torch.equal(student.encoder.embeddings, teacher.encoder.embeddings)


def test_distillation_finetune_step(oracle_wrappers):
    """one global fine-tuning step: teacher-rendered RGB + depth targets through the proxy, MSE + L1(depth)"""
    from nerf import synthetic as syn
    from sealnerf import SealTrainer
    teacher, student, mapper = _nets(oracle_wrappers)
    tr = SealTrainer(student, teacher, lr=1e-2, fp16=False)
    poses = syn.orbit_poses(1, seed=0)
    r = syn.get_rays(poses, syn.lego_intrinsics(32, 32), 32, 32)
    ro, rd = r["rays_o"], r["rays_d"]
    gt_rgb, gt_depth = tr.proxy_truth(ro, rd)
    assert gt_rgb.shape == (1, 1024, 3) and gt_depth.shape == (1, 1024) and torch.isfinite(gt_rgb).all()
    assert (gt_depth > 0).any(), "force-filled edit region must be sampled by the teacher"
    student.hack_bitfield()
    student.mean_count = 0
    before = student.encoder.embeddings.detach().clone()
    loss = tr.train_step(ro, rd, gt_rgb, gt_depth)
    assert torch.isfinite(loss) and not torch.equal(before, student.encoder.embeddings)


_DP = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["S3D_REPO"]); sys.path.insert(0, os.path.join(os.environ["S3D_REPO"], "seal-3d_amd"))
sys.path.insert(0, os.path.join(os.environ["S3D_REPO"], "tests"))
from oracle import oracle_backend as ob
import gridencoder.grid as gg, shencoder.sphere_harmonics as sh, raymarching.raymarching as rm
gg._backend, sh._backend, rm._backend = ob.GridBackend, ob.SHBackend, ob.RaymarchingBackend
ob.set_threads(2)
from parallel import RayShardedDP, init_from_env
from sealnerf import SealTrainer
import types, test_seal
rank, world, _ = init_from_env("gloo")
teacher, student, mapper = test_seal._nets(types.SimpleNamespace())
dp = RayShardedDP() if world > 1 else None
tr = SealTrainer(student, teacher, lr=1e-2, fp16=False, dist=dp)
tr.init_pretraining(batch_size=100000, lr=0.05, local_point_step=0.05)
loss = float(tr.pretrain_one_epoch())
emb = student.encoder.embeddings.detach().clone()
# two fine-tuning steps, each rank on its own rays, occupancy update before every step (ADVICE r1: the replicas' grids
# must stay identical — the sweep is sharded over the ranks with a common random stream)
from nerf import synthetic as syn
tr.update_extra_interval = 1
torch.manual_seed(50 + rank)
r = syn.get_rays(syn.orbit_poses(2, seed=1)[rank:rank + 1], syn.lego_intrinsics(24, 24), 24, 24)
for _ in range(2):
    ft = float(tr.train_step(r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()))
torch.save({"loss": loss, "emb": emb, "bits": student.density_bitfield.clone(), "grid": student.density_grid.clone(),
            "emb_ft": student.encoder.embeddings.detach().clone(), "mean_count": student.mean_count, "ft": ft},
           os.environ["S3D_OUT"] + f".{world}.{rank}")
if world > 1: dist.destroy_process_group()
'''


def test_sharded_pretraining_equals_single_process(tmp_path):
    script = tmp_path / "dp.py"
    script.write_text(_DP)
    out = str(tmp_path / "res")
    env = dict(os.environ, S3D_REPO=REPO, S3D_OUT=out, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r1 = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=400)
    assert r1.returncode == 0, r1.stderr[-2000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a = torch.load(out + ".1.0")
    b0, b1 = torch.load(out + ".2.0"), torch.load(out + ".2.1")
    assert torch.equal(b0["emb"], b1["emb"]), "replicas diverged"
    torch.testing.assert_close(b0["emb"], a["emb"], rtol=1e-4, atol=1e-6)   # Adam step of summed shard grads == full grad
    assert abs(b0["loss"] + b1["loss"] - a["loss"]) < 1e-4 * max(1.0, abs(a["loss"]))
    # after fine-tuning steps with occupancy updates: identical occupancy state and identical weights on both replicas
    assert torch.equal(b0["bits"], b1["bits"]) and torch.equal(b0["grid"], b1["grid"]) and b0["mean_count"] == b1["mean_count"]
    assert torch.equal(b0["emb_ft"], b1["emb_ft"]) and not torch.equal(b0["emb_ft"], b0["emb"])
    assert int(b0["bits"].sum()) > 0 and b0["ft"] == b0["ft"]
