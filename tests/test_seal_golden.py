"""CPU: the build's bbox proxy mapper (sealnerf/seal_utils.py, torch op sequence) against tests/golden/seal_bbox.npz — the
outputs of the REFERENCE's `SealBBoxMapper.map_to_origin` / `map_mask` / `points_in_mesh` / `moller_trumbore`
(SealNeRF/seal_utils.py:132-153, 237-279, 630-685) executed on the same seeded points (oracle/gen_golden.py seal).
What the fixture does NOT pin: the construction of the box meshes (trimesh's oriented bounding box is absent here); the
triangles and bounds the reference functions ran on are the build's and are stored in the fixture."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def case_config(tag, S):
    cfg = {"type": "bbox", "raw": S[f"{tag}_raw"].tolist(), "transform": S[f"{tag}_transform"].tolist(),
           "scale": S[f"{tag}_scale"].tolist(), "boundType": str(S[f"{tag}_bound_type"])}
    if S[f"{tag}_map_source"].size:
        cfg["mapSource"] = S[f"{tag}_map_source"].tolist()
    return cfg


@pytest.fixture(scope="module")
def S():
    return np.load(os.path.join(GOLDEN, "seal_bbox.npz"))


@pytest.mark.parametrize("tag", ["both", "to", "from_rot"])
def test_torch_mapper_matches_reference_execution(S, tag):
    from sealnerf import SealBBoxMapper
    mapper = SealBBoxMapper(case_config(tag, S))
    # the constants the reference functions ran on are reproduced by the constructor (deterministic host arithmetic)
    assert np.array_equal(mapper.map_triangles.numpy(), S[f"{tag}_triangles"])
    assert np.array_equal(mapper.map_data["map_bound"].numpy(), S[f"{tag}_map_bound"])
    pts, dirs = torch.from_numpy(S[f"{tag}_points"]), torch.from_numpy(S[f"{tag}_dirs"])
    p, d, m = mapper.map_to_origin(pts, dirs)
    assert torch.equal(m, torch.from_numpy(S[f"{tag}_mask"]))
    assert 500 < int(m.sum()) < 5000
    # same torch op sequence on the same CPU: bit-exact
    assert np.array_equal(p.numpy(), S[f"{tag}_out_points"]) and np.array_equal(d.numpy(), S[f"{tag}_out_dirs"])


def test_rotated_raw_points_give_an_oriented_source_box(S):
    """`raw` spanning a rotated box must give that (oriented) box, not its axis-aligned hull"""
    from sealnerf.seal_utils import oriented_box_vertices
    raw = S["from_rot_raw"]
    v = oriented_box_vertices(raw)
    d = np.abs(v[:, None] - raw[None]).sum(-1).min(1)
    assert d.max() < 1e-12  # the eight corners are the raw points themselves
    lo, hi = raw.min(0), raw.max(0)
    assert np.prod(hi - lo) > 1.2 * 0.4 * 0.3 * 0.4  # (the axis-aligned hull is visibly larger)


@pytest.mark.parametrize("name", ["hsv", "rgb", "both"])
def test_map_color_matches_reference_execution(S, name):
    """colour remapping of the bbox tool (seal_utils.py:48-58, 739-769; color_utils.py:33-66): the reference's `map_color`
    executed on 4,000 seeded colours (greys, pure channels, channel ties, hue wrap-around) vs the build's closed-form torch
    twin — the reference assigns through boolean masks, the twin selects with `where`/`gather`: same values to 1 ulp."""
    from sealnerf import SealBBoxMapper
    opts = S[f"color_{name}_opts"]
    cfg = case_config("to", S)
    if not np.isnan(opts[0]).any():
        cfg["hsv"] = opts[0].tolist()
    if not np.isnan(opts[1]).any():
        cfg["rgb"] = opts[1].tolist()
        cfg["rgbLightOffset"] = float(opts[2, 0])
    mapper = SealBBoxMapper(cfg)
    cols = torch.from_numpy(S["color_in"])
    got = mapper.map_color(None, None, cols.clone())
    want = torch.from_numpy(S[f"color_{name}"])
    assert got.shape == want.shape and torch.isfinite(got).all()
    torch.testing.assert_close(got, want, rtol=0, atol=2e-7)
    assert not torch.allclose(got, cols, atol=1e-3), "the remap must change the colours"
    # identity without options (the BASELINE configs)
    plain = SealBBoxMapper(case_config("to", S))
    assert plain.map_color(None, None, cols) is cols
