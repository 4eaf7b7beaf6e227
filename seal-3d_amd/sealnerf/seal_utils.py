"""Seal proxy functions (SealNeRF/seal_utils.py): map edited-space points back to the source space.

Only the bounding-box tool of BASELINE configs 3/4 is implemented (`SealBBoxMapper`, seal_utils.py:155-279):
config keys `type: bbox`, `raw` (points spanning the source box), `transform` (4x4 source->target), `scale` (3),
`boundType` ('to' | 'from' | 'both'), optional `mapSource`.  No trimesh / pytorch3d: the box meshes are built
directly (12 triangles per box) and the inside test is the reference's two-ray Moller-Trumbore parity test
(seal_utils.py:630-685) in plain torch.  Colour remapping: the bbox tool's `hsv` / `rgb` options (seal_utils.py:48-58,
739-769, color_utils.py:33-66); the brush tool's image remap is not part of the bbox configuration.
"""
import json

import numpy as np
import torch

_BOX_FACES = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
                       [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]])
# fixed test direction of the reference's inside test (seal_utils.py:676-678)
_TEST_DIR = (0.4395064455, 0.617598629942, 0.652231566745)


def _box_vertices(lo, hi):
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], dtype=np.float64)


def _min_area_rect(p2):
    """minimum-area enclosing rectangle of 2-D points: (area, angle of the rectangle's first axis); one side of the optimum
    is collinear with a hull edge, so the hull edges are the only candidates"""
    from scipy.spatial import ConvexHull
    hull = p2[ConvexHull(p2).vertices]
    best = (np.inf, 0.0)
    for a, b in zip(hull, np.roll(hull, -1, axis=0)):
        e = b - a
        n = np.linalg.norm(e)
        if n < 1e-12:
            continue
        e = e / n
        q = p2 @ np.stack([e, [-e[1], e[0]]], axis=1)
        area = np.prod(q.max(0) - q.min(0))
        if area < best[0]:
            best = (area, np.arctan2(e[1], e[0]))
    return best


def oriented_box_vertices(points, snap=1e-9):
    """8 corners of the minimum-volume oriented bounding box of `points` — what the reference gets from
    `trimesh.PointCloud(raw).bounding_box_oriented` (seal_utils.py:587-588; trimesh.bounds.oriented_bounds: one face of
    the optimum is parallel to a convex-hull facet, so every facet normal is tried with the minimum-area rectangle of the
    projection).  Corner order is that of `_box_vertices` (index bits = side along the box's own x, y, z axes), so
    `_BOX_FACES` triangulates it.  A box whose axes are the coordinate axes (within `snap`) is returned as the exact
    min / max corners; coplanar or fewer than 4 points fall back to the axis-aligned box."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    aabb = _box_vertices(lo, hi)
    try:
        from scipy.spatial import ConvexHull
        hull = ConvexHull(pts)
    except Exception:  # degenerate input (QhullError) or no scipy: the axis-aligned box
        return aabb
    hp = pts[hull.vertices]
    best = None
    seen = []
    for eq in hull.equations:
        n = eq[:3] / np.linalg.norm(eq[:3])
        if n[np.argmax(np.abs(n))] < 0:
            n = -n  # a normal and its opposite give the same box
        if any(abs(abs(n @ m) - 1) < 1e-12 for m in seen):
            continue
        seen.append(n)
        ref = np.eye(3)[np.argmin(np.abs(n))]
        u = np.cross(n, ref)
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        area, ang = _min_area_rect(hp @ np.stack([u, v], axis=1))
        h = hp @ n
        vol = area * (h.max() - h.min())
        if best is None or vol < best[0] * (1 - 1e-12):
            c, s_ = np.cos(ang), np.sin(ang)
            best = (vol, np.stack([c * u + s_ * v, -s_ * u + c * v, n]))  # rows = box axes
    vol_aabb = np.prod(hi - lo)
    R = best[1]
    # the box's axes up to order / sign are the coordinate axes -> the exact AABB (also when it is not smaller)
    if best[0] >= vol_aabb * (1 - 1e-12) or np.all(np.abs(np.abs(R).max(1) - 1) < snap):
        return aabb
    # canonical axis order / sign: each box axis is assigned to the coordinate axis it is closest to, pointing along +
    order = []
    for k in range(3):
        cand = [i for i in range(3) if i not in order]
        order.append(max(cand, key=lambda i: abs(R[i, k])))
    R = R[order]
    R = R * np.where(R[np.arange(3), np.arange(3)] < 0, -1.0, 1.0)[:, None]
    q = pts @ R.T
    qlo, qhi = q.min(0), q.max(0)
    return _box_vertices(qlo, qhi) @ R


def moller_trumbore_any(ray_o, ray_d, tris, eps=1e-8):
    """does ray i hit any triangle?  (n_rays, 3), (n_rays, 3), (n_faces, 3, 3)  — seal_utils.py:630-665"""
    E1 = tris[:, 1] - tris[:, 0]
    E2 = tris[:, 2] - tris[:, 0]
    N = torch.cross(E1, E2, dim=-1)
    invdet = 1.0 / -(torch.einsum("md,nd->mn", ray_d, N) + eps)
    A0 = ray_o[:, None] - tris[None, :, 0]
    DA0 = torch.cross(A0, ray_d[:, None].expand(*A0.shape), dim=-1)
    u = torch.einsum("mnd,nd->mn", DA0, E2) * invdet
    v = -torch.einsum("mnd,nd->mn", DA0, E1) * invdet
    t = torch.einsum("mnd,nd->mn", A0, N) * invdet
    return ((t >= 0.0) & (u >= 0.0) & (v >= 0.0) & ((u + v) <= 1.0)).any(1)


def points_in_mesh(points, triangles):
    """a point is inside iff rays in BOTH directions of the test axis hit the mesh (seal_utils.py:668-685)"""
    d = torch.tensor([_TEST_DIR], device=points.device, dtype=points.dtype).repeat(points.shape[0], 1)
    hit = moller_trumbore_any(torch.cat([points, points]), torch.cat([d, -d]), triangles)
    return hit[:points.shape[0]] & hit[points.shape[0]:]


def rgb_to_hsv(rgb):
    """color_utils.py:33-46 (`rgb2hsv_torch`) on [N, 3], closed form instead of boolean-mask scatters: hue from the FIRST
    maximal channel (torch.max's tie rule), `%` = floored modulo, grey (delta == 0) -> hue 0; no host sync"""
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    cmax, idx = torch.max(rgb, dim=1)
    cmin = torch.min(rgb, dim=1)[0]
    delta = cmax - cmin
    safe = torch.where(delta == 0, torch.ones_like(delta), delta)
    h0 = torch.remainder((g - b) / safe, 6.0)
    h1 = (b - r) / safe + 2.0
    h2 = (r - g) / safe + 4.0
    h = torch.where(idx == 0, h0, torch.where(idx == 1, h1, h2))
    h = torch.where(delta == 0, torch.zeros_like(h), h) / 6.0
    s = torch.where(cmax == 0, torch.zeros_like(cmax), delta / torch.where(cmax == 0, torch.ones_like(cmax), cmax))
    return torch.stack([h, s, cmax], dim=1)


def hsv_to_rgb(hsv):
    """color_utils.py:49-66 (`hsv2rgb_torch`) on [N, 3]; the sextant is `(h * 6)` cast to uint8, modulo 6, as there"""
    h, s, v = hsv[:, 0], hsv[:, 1], hsv[:, 2]
    c = v * s
    x = c * (-torch.abs(torch.remainder(h * 6.0, 2.0) - 1.0) + 1.0)
    m = v - c
    o = torch.zeros_like(c)
    idx = (h * 6.0).to(torch.uint8) % 6
    table = torch.stack([torch.stack([c, x, o], 1), torch.stack([x, c, o], 1), torch.stack([o, c, x], 1),
                         torch.stack([o, x, c], 1), torch.stack([x, o, c], 1), torch.stack([c, o, x], 1)], dim=1)  # [N, 6, 3]
    rgb = table.gather(1, idx.long()[:, None, None].expand(-1, 1, 3))[:, 0]
    return rgb + m[:, None]


def modify_hsv(rgb, modification):
    """seal_utils.py:739-750: rgb -> hsv, add the offsets, -> rgb"""
    if rgb.shape[0] == 0:
        return rgb
    hsv = rgb_to_hsv(rgb)
    mod = torch.as_tensor(modification, dtype=rgb.dtype, device=rgb.device)
    return hsv_to_rgb(hsv + mod[None, :3])


def modify_rgb(rgb, modification, light_offset=0):
    """seal_utils.py:753-769: hue and saturation of the target colour, value = the target's value + the sample's offset from
    the batch's mean value (+ light_offset), clamped to [0, 1]"""
    if rgb.shape[0] == 0:
        return rgb
    hsv = rgb_to_hsv(rgb)
    mod = rgb_to_hsv(torch.as_tensor(modification, dtype=rgb.dtype, device=rgb.device).view(-1, 3))
    raw = hsv[:, 2]
    val = torch.clamp(mod[:, 2] + (raw - raw.mean()) + light_offset, 0.0, 1.0)
    out = torch.stack([mod[:, 0].expand_as(val), mod[:, 1].expand_as(val), val], dim=1)
    return hsv_to_rgb(out)


class SealBBoxMapper:
    def __init__(self, seal_config):
        self.config = seal_config
        T = np.array(seal_config["transform"], dtype=np.float64)
        scale = np.array(seal_config["scale"], dtype=np.float64)
        raw = np.array(seal_config["raw"], dtype=np.float64)
        # the reference's `get_trimesh_box(raw)`: the ORIENTED bounding box of the raw points (seal_utils.py:186-188, 587-588)
        from_v = oriented_box_vertices(raw)
        center = from_v.mean(0)
        to_v = (from_v - center) * scale + center
        to_v = to_v @ T[:3, :3].T + T[:3, 3]
        to_center = to_v.mean(0)
        from_b = np.stack([from_v.min(0), from_v.max(0)])
        to_b = np.stack([to_v.min(0), to_v.max(0)])
        fill = np.stack([to_b, from_b])  # [2 boxes, (min,max), 3]  == force_fill_bound
        kind = seal_config.get("boundType", "to")
        if kind == "to":
            bounds, verts = to_b, [to_v]
        elif kind == "from":
            bounds, verts = from_b, [from_v]
        elif kind == "both":
            bounds, verts = fill, [to_v, from_v]
        else:
            raise ValueError(f"unknown boundType {kind}")
        tris = np.concatenate([v[_BOX_FACES] for v in verts])
        self.map_data = {
            "force_fill_bound": torch.tensor(fill, dtype=torch.float32),
            "map_bound": torch.tensor(bounds, dtype=torch.float32),
            "pose_center": torch.tensor((center + to_center) / 2, dtype=torch.float32),
            "pose_radius": float(np.linalg.norm(center - to_center) * 10),
            "transform": torch.tensor(np.linalg.inv(T), dtype=torch.float32),
            "rotation": torch.tensor(np.linalg.inv(T[:3, :3]), dtype=torch.float32),
            "scale": torch.tensor(1.0 / scale, dtype=torch.float32),
            "center": torch.tensor(center, dtype=torch.float32),
        }
        if "hsv" in seal_config:  # seal_utils.py:226-230
            self.map_data["hsv"] = torch.tensor(seal_config["hsv"], dtype=torch.float32)
        if "rgb" in seal_config:
            self.map_data["rgb"] = torch.tensor(seal_config["rgb"], dtype=torch.float32)
            self.map_data["rgb_light_offset"] = float(seal_config.get("rgbLightOffset", 0))
        if seal_config.get("mapSource"):
            self.map_data["empty_bound"] = torch.tensor(from_b, dtype=torch.float32)
            self.map_data["map_source"] = torch.tensor(seal_config["mapSource"], dtype=torch.float32)
        self.map_triangles = torch.tensor(tris, dtype=torch.float32)
        self.device = torch.device("cpu")
        # the same constants as float32 host arrays for the device kernel (csrc/seal.hip)
        md = self.map_data
        self._host = {"triangles": self.map_triangles.numpy().copy(),
                      "bounds": md["map_bound"].numpy().reshape(-1, 2, 3).copy(),
                      "inv_transform": md["transform"].numpy().copy(), "inv_rotation": md["rotation"].numpy().copy(),
                      "inv_scale": md["scale"].numpy().copy(), "center": md["center"].numpy().copy()}
        if "map_source" in md:
            self._host["empty_bound"] = md["empty_bound"].numpy().copy()
            self._host["map_source"] = md["map_source"].numpy().copy()

    def to(self, device):
        if device != self.device:
            self.map_data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.map_data.items()}
            self.map_triangles = self.map_triangles.to(device)
            self.device = device
        return self

    def map_mask(self, points):
        """AABB pre-test (incl. the reference's `points.all(1)` term) then the mesh inside test — seal_utils.py:132-153"""
        bounds = self.map_data["map_bound"]
        if bounds.ndim == 2:
            bounds = bounds[None]
        mask = None
        for i in range(bounds.shape[0]):
            cur = points.all(1) & ((bounds[i][1] > points) & (points > bounds[i][0])).all(1)
            mask = cur if mask is None else (mask | cur)
        if not mask.any():
            return mask
        inside = points_in_mesh(points[mask], self.map_triangles)
        mask[mask.clone()] = inside
        return mask

    native = True  # GPU tensors go through the one-pass device kernel; False = the reference's torch op sequence

    def _map_native(self, points, dirs):
        import s3d_hip
        lead = points.shape
        p = points.reshape(-1, 3).contiguous()
        d = dirs.reshape(-1, 3).float().contiguous() if dirs is not None else None
        out_p = torch.empty_like(p)
        out_d = torch.empty_like(d) if d is not None else None
        mask = torch.empty(p.shape[0], dtype=torch.bool, device=p.device)
        # inside a renderer's announced padded batch (s3d_hip.row_limit) the rows behind the sample count are skipped, like
        # in every other per-sample kernel of that path
        s3d_hip.SealBackend.bbox_map(p, d, self._host, out_p, out_d, mask.view(torch.uint8), s3d_hip.active_row_limit(p.shape[0]))
        return out_p.view(lead), (out_d.view(dirs.shape) if dirs is not None else None), mask

    @torch.autocast("cuda", enabled=False)
    def map_to_origin(self, points, dirs=None):
        """seal_utils.py:237-279"""
        if self.native and points.is_cuda and points.dtype == torch.float32 and points.shape[-1] == 3 and points.numel() > 0:
            return self._map_native(points, dirs)
        self.to(points.device)
        mask = self.map_mask(points)
        if not mask.any():
            return points, dirs, mask
        inner = points[mask]
        hom = torch.vstack([inner.T, torch.ones([1, inner.shape[0]], device=inner.device)])
        moved = torch.matmul(self.map_data["transform"], hom).T[:, :3]
        origin = (moved - self.map_data["center"]) * self.map_data["scale"] + self.map_data["center"]
        out_p = points.clone()
        out_d = dirs.clone() if dirs is not None else None
        if "map_source" in self.map_data:
            sb = self.map_data["empty_bound"]
            out_p[((sb[1] > points) & (points > sb[0])).all(1)] = self.map_data["map_source"]
        out_p[mask] = origin
        if dirs is not None:
            out_d[mask] = torch.matmul(self.map_data["rotation"], dirs[mask].T).T
        return out_p, out_d, mask

    def map_color(self, points, dirs, colors):
        """seal_utils.py:48-81 for the bbox tool (`hsv` / `rgb` of seal.json, :226-230): hue / saturation / value offsets, then
        re-colouring towards a target RGB that keeps each sample's brightness offset from the batch mean.  The image remap
        (`image`) belongs to the brush tool and is not part of the bbox configuration."""
        if "hsv" in self.map_data:
            colors = modify_hsv(colors, self.map_data["hsv"])
        if "rgb" in self.map_data:
            colors = modify_rgb(colors, self.map_data["rgb"], self.map_data.get("rgb_light_offset", 0))
        return colors


    def map_color_masked(self, points, dirs, colors, mask):
        """the renderers' use of map_color (SealNeRF/renderer.py:316, 396-399): `colors[mask] = map_color(points[mask],
        dirs[mask], colors[mask])` — returns a new tensor, `colors` is left alone"""
        md = self.map_data
        if (self.native and colors.is_cuda and mask is not None and colors.dtype in (torch.float32, torch.float16) and colors.dim() == 2
                and colors.shape[1] == 3 and "image" not in md and ("hsv" in md or "rgb" in md)):
            # one or two passes on the device (csrc/seal.hip: s3d_seal_map_color) instead of a boolean gather (host sync), ~40
            # masked elementwise launches and a scatter back; the batch mean of the `rgb` edit is an order-independent sum
            import s3d_hip
            src = colors.contiguous()
            out = torch.empty_like(src)
            hsv = md["hsv"].tolist() if "hsv" in md else None
            tgt = md["rgb"].tolist() if "rgb" in md else None
            s3d_hip.SealBackend.map_color(src, mask.view(torch.uint8), hsv, tgt, md.get("rgb_light_offset", 0) if tgt is not None else 0.0,
                                          out, n_valid=s3d_hip.active_row_limit(src.shape[0]))
            return out
        out = colors.clone()
        if mask is None:
            return self.map_color(points, dirs, out)
        sel = colors[mask]
        if sel.shape[0]:
            out[mask] = self.map_color(points[mask] if points is not None else None, dirs[mask] if dirs is not None else None,
                                       sel.float()).to(colors.dtype)
        return out


def get_seal_mapper(config_dict=None, config_file=None):
    """seal_utils.py:573-584 (plain JSON instead of json5)"""
    if config_dict is None:
        with open(config_file) as f:
            config_dict = json.load(f)
    if config_dict["type"] == "bbox":
        return SealBBoxMapper(config_dict)
    raise NotImplementedError(f"seal tool `{config_dict['type']}` is outside the BASELINE configs (bbox only)")
