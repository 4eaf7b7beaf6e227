"""Synthetic "lego-like" inputs for tests and benchmarks.

There is no dataset in the build/bench environment, so workloads are generated:
  * a procedural scene: the union of ~40 seeded axis-aligned boxes inside
    [-0.7, 0.7]^3 rasterised onto the 128^3 occupancy grid (morton order),
  * Blender-Lego cameras: 800x800, camera_angle_x = 0.6911112
    (fl = 400 / tan(0.3456) = 1111.11), poses on the upper hemisphere at radius
    4.0311 * 0.8 looking at the origin, built with the orbit-camera convention
    of nerf/provider.py:57-91 (`rand_poses`),
  * rays with the pixel-centre convention of nerf/utils.py:124-133 (`get_rays`).
Everything is seeded and device-agnostic (CPU for the oracle, GPU for the product).
"""
import math

import numpy as np
import torch

LEGO_W = LEGO_H = 800
LEGO_ANGLE_X = 0.6911112
LEGO_RADIUS = 4.0311 * 0.8


def lego_intrinsics(H=LEGO_H, W=LEGO_W):
    fl = W / (2 * math.tan(LEGO_ANGLE_X / 2))
    return np.array([fl, fl, W / 2, H / 2], dtype=np.float64)


def _normalize(v):
    return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-10)


def orbit_poses(n, seed=0, radius=LEGO_RADIUS, theta_range=(math.pi / 6, math.pi / 2 - 0.05), device="cpu"):
    """cam2world matrices on the upper hemisphere, look-at origin (convention of `rand_poses`)."""
    g = torch.Generator().manual_seed(seed)
    thetas = torch.rand(n, generator=g) * (theta_range[1] - theta_range[0]) + theta_range[0]
    phis = torch.rand(n, generator=g) * 2 * math.pi
    centers = torch.stack([radius * torch.sin(thetas) * torch.sin(phis),
                           radius * torch.cos(thetas),
                           radius * torch.sin(thetas) * torch.cos(phis)], dim=-1)
    fwd = -_normalize(centers)
    up = torch.tensor([0.0, -1.0, 0.0]).expand(n, 3)
    right = _normalize(torch.cross(fwd, up, dim=-1))
    up = _normalize(torch.cross(right, fwd, dim=-1))
    poses = torch.eye(4).repeat(n, 1, 1)
    poses[:, :3, :3] = torch.stack((right, up, fwd), dim=-1)
    poses[:, :3, 3] = centers
    return poses.to(device)


def get_rays(poses, intrinsics, H, W, N=-1, generator=None):
    """Pixel-centre rays (nerf/utils.py:54-139 without error maps / patches).
    poses [B,4,4] cam2world; returns dict(rays_o [B,n,3], rays_d [B,n,3], inds [B,n])."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    if N > 0:
        N = min(N, H * W)
        inds = torch.randint(0, H * W, size=[N], generator=generator).to(device)
        inds = inds.expand([B, N])
    else:
        inds = torch.arange(H * W, device=device).expand([B, H * W])
    i = (inds % W).float() + 0.5
    j = (inds // W).float() + 0.5
    zs = torch.ones_like(i)
    dirs = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    rays_d = dirs @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return {"rays_o": rays_o, "rays_d": rays_d, "inds": inds}


def _expand_bits(v):
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v


def morton3d_np(x, y, z):
    x, y, z = (np.asarray(a, dtype=np.uint64) for a in (x, y, z))
    return (_expand_bits(x) | (_expand_bits(y) << np.uint64(1)) | (_expand_bits(z) << np.uint64(2))).astype(np.int64)


def lego_like_boxes(seed=0, n_boxes=40):
    rng = np.random.RandomState(seed)
    ctr = rng.uniform(-0.5, 0.5, size=(n_boxes, 3))
    half = rng.uniform(0.04, 0.2, size=(n_boxes, 3))
    lo = np.clip(ctr - half, -0.7, 0.7)
    hi = np.clip(ctr + half, -0.7, 0.7)
    return lo.astype(np.float32), hi.astype(np.float32)


def lego_like_density_grid(seed=0, n_boxes=40, H=128, cascade=1, bound=1.0, sigma=50.0):
    """density grid [cascade, H^3] (morton order) for the box scene, plus the packed bitfield (numpy)."""
    lo, hi = lego_like_boxes(seed, n_boxes)
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    idx = np.arange(H)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    mort = morton3d_np(X.ravel(), Y.ravel(), Z.ravel())
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        # cell centres of cascade `cas` in world units
        cx = ((X.ravel() + 0.5) / H * 2 - 1) * b
        cy = ((Y.ravel() + 0.5) / H * 2 - 1) * b
        cz = ((Z.ravel() + 0.5) / H * 2 - 1) * b
        occ = np.zeros(H ** 3, dtype=bool)
        for k in range(lo.shape[0]):
            occ |= ((cx >= lo[k, 0]) & (cx <= hi[k, 0]) & (cy >= lo[k, 1]) & (cy <= hi[k, 1]) &
                    (cz >= lo[k, 2]) & (cz <= hi[k, 2]))
        grid[cas, mort] = np.where(occ, sigma, 0.0)
    bits = np.packbits((grid.reshape(-1) > 0.01).astype(np.uint8), bitorder="little")
    return grid, bits


def box_density(xyz, lo, hi, sigma=50.0):
    """Analytic density of the box scene at world points (torch, [..., 3])."""
    lo_t = torch.as_tensor(lo, device=xyz.device)
    hi_t = torch.as_tensor(hi, device=xyz.device)
    inside = ((xyz[..., None, :] >= lo_t) & (xyz[..., None, :] <= hi_t)).all(-1).any(-1)
    return inside.to(xyz.dtype) * sigma
