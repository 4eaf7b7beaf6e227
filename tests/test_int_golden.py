"""The integer kernels against the REFERENCE TEXT (tests/golden/int_kernels.npz).

`oracle/gen_golden.py int` transliterates raymarching.cu:42-81 (`mip_from_pos`, `mip_from_dt`, `__expand_bits`,
`__morton3D`, `__morton3D_invert`), gridencoder.cu:50-84 (`fast_hash`, `get_grid_index`) and the index lines of
`kernel_grid` (:137-139, 148-149, 165-180) statement by statement and evaluates them with numpy uint32 / int32 / float32
semantics on seeded inputs.  Here the CPU oracle must reproduce every value bit for bit (the HIP kernels:
tests/test_gpu_golden.py) — morton codes, cascade selection and table rows are then pinned to the reference's own
source text, not to a second restatement of ours."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "int_kernels.npz"))


def _morton(ob, coords):
    c = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.int32))
    out = torch.empty(c.shape[0], dtype=torch.int32)
    ob.RaymarchingBackend.morton3D(c, c.shape[0], out)
    return out.numpy()


def _invert(ob, ind):
    i = torch.from_numpy(np.ascontiguousarray(ind, dtype=np.int32))
    out = torch.empty(i.shape[0], 3, dtype=torch.int32)
    ob.RaymarchingBackend.morton3D_invert(i, i.shape[0], out)
    return out.numpy()


def test_expand_bits_and_morton(oracle, G):
    v = G["expand_in"]
    lo = v < 1024       # `__morton3D(x, 0, 0) == __expand_bits(x)`; the oracle exports the kernel, not the helper
    c = np.zeros((int(lo.sum()), 3), np.int32)
    c[:, 0] = v[lo]
    assert np.array_equal(_morton(oracle, c).view(np.uint32), G["expand_out"][lo])
    for axis in (1, 2):
        c2 = np.zeros_like(c)
        c2[:, axis] = v[lo]
        assert np.array_equal(_morton(oracle, c2).view(np.uint32), G["expand_out"][lo] << axis)
    # inputs above 10 bits: the helper's wrap-around arithmetic shows through the x slot of the kernel
    c = np.zeros((v.size, 3), np.int32)
    c[:, 0] = v.view(np.int32)
    assert np.array_equal(_morton(oracle, c).view(np.uint32), G["expand_out"])
    assert np.array_equal(_morton(oracle, G["morton_coords"]), G["morton_indices"])


def test_morton_full_sweep_checksum(oracle, G):
    g = np.arange(128, dtype=np.int32)
    sweep = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    sw = _morton(oracle, sweep)
    assert np.uint32(zlib.crc32(sw.tobytes())) == G["morton_sweep128_crc"]
    inv = _invert(oracle, np.arange(128 ** 3, dtype=np.int32))
    assert np.uint32(zlib.crc32(inv.tobytes())) == G["invert_sweep128_crc"]


def test_morton_invert_including_negative_indices(oracle, G):
    assert np.array_equal(_invert(oracle, G["invert_indices"]), G["invert_coords"])


@pytest.mark.parametrize("C", [1, 2, 4, 8])
def test_cascade_selection(oracle, G, C):
    assert np.array_equal(oracle.mip_from_pos(G["mip_xyz"], C), G[f"mip_pos_c{C}"])
    assert np.array_equal(oracle.mip_from_dt(G["mip_dt"], 128, C), G[f"mip_dt_c{C}"])


def test_grid_index_on_raw_cell_coordinates(oracle, G):
    cases = G["raw_cases"]
    assert len(cases) == 64
    for k, (D, C, gridtype, ac, ch, hs, res) in enumerate(cases.tolist()):
        pg = G[f"raw_d{D}_pos_grid"]
        got = np.array([oracle.grid_index(D, C, gridtype, ac, ch, hs, res, row) for row in pg], np.uint32)
        assert np.array_equal(got, G[f"raw_idx_{k}"]), (k, D, C, gridtype, ac, ch, hs, res)


@pytest.mark.parametrize("D", [2, 3])
def test_fast_hash_through_a_hashed_level(oracle, G, D):
    """`fast_hash` itself: a level whose stride exceeds the table on the first axis takes the hash for every cell, and a
    2^32-row table (hashmap_size 0 is not a table; 2^31 keeps the top bit out) leaves it visible up to the modulo."""
    pg = G[f"raw_d{D}_pos_grid"]
    hs = 1 << 31
    got = np.array([oracle.grid_index(D, 1, 0, 0, 0, hs, 0xFFFFFFFE, row) for row in pg], np.uint32)
    # resolution + 1 = 2^32 - 1 > hs after the first axis -> hashed; rows = hash % 2^31
    assert np.array_equal(got, G[f"raw_d{D}_fast_hash"] % np.uint32(hs))


@pytest.mark.parametrize("tag", ["lego", "hash", "smooth", "tiled_ac"])
def test_level_table_cells_and_corner_rows(oracle, G, tag):
    D, C, gridtype, ac, L, H = G[f"grid_{tag}_cfg"].tolist()
    S = float(G[f"grid_{tag}_S"])
    scales = oracle.level_scales(L, S, H)
    assert np.array_equal(scales, G[f"grid_{tag}_scales"])
    assert np.array_equal(np.ceil(scales).astype(np.int64) + 1, G[f"grid_{tag}_resolution"])
    x = torch.from_numpy(G[f"grid_{tag}_x"])
    offsets = torch.from_numpy(G[f"grid_{tag}_offsets"])
    B = x.shape[0]
    emb = torch.zeros(int(offsets[-1]), C)
    out = torch.empty(L, B, C)
    cidx = torch.empty(B, L, 1 << D, dtype=torch.int32)
    oracle.lib().s3o_grid_encode_forward(oracle._p(x), oracle._p(emb), oracle._p(offsets), oracle._p(out), oracle._u(B), oracle._u(D),
                                         oracle._u(C), oracle._u(L), oracle._p(scales), None, oracle._u(gridtype),
                                         int(ac), oracle._u(0), 0, oracle._p(cidx))
    rows = cidx.numpy().view(np.uint32)
    assert np.array_equal(rows.astype(np.int64) * C, G[f"grid_{tag}_index"].astype(np.int64))
    if gridtype == 0 and not ac:      # a dense level's corner-0 row decodes to the cell: compare with the fixture's cells
        res = G[f"grid_{tag}_resolution"]
        hs = np.diff(G[f"grid_{tag}_offsets"])
        for l in range(L):
            if (res[l] + 1) ** D <= hs[l]:
                r0 = rows[:, l, 0].astype(np.int64)
                cell = np.stack([(r0 // (res[l] + 1) ** d) % (res[l] + 1) for d in range(D)], -1)
                assert np.array_equal(cell, G[f"grid_{tag}_pos_grid"][:, l].astype(np.int64)), l


def test_packbits_from_the_reference_expression(oracle, G):
    """kernel_packbits (raymarching.cu:262-289): `bits |= (grid[i] > density_thresh) ? ((uint8_t)1 << i) : 0` — the comparison
    incl. the threshold itself, its neighbours, +-0, +-inf and NaN"""
    cells = G["packbits_grid"]
    grid = torch.from_numpy(np.ascontiguousarray(cells.reshape(-1)))
    for th in (10.0, 0.0, 0.01):
        bf = torch.zeros(cells.shape[0], dtype=torch.uint8)
        oracle.RaymarchingBackend.packbits(grid, cells.shape[0], th, bf)
        assert np.array_equal(bf.numpy(), G[f"packbits_thresh{th:g}"]), th


def test_near_far_from_the_reference_statements(oracle, G):
    """kernel_near_far_from_aabb (raymarching.cu:92-145) evaluated ray by ray: hits, misses (FLT_MAX), axis-parallel rays
    (reciprocal +-inf), origins on a slab plane (0 x inf = NaN in a comparison), min_near clamp — bit for bit"""
    ro, rd = torch.from_numpy(G["nearfar_rays_o"]), torch.from_numpy(G["nearfar_rays_d"])
    aabb = torch.from_numpy(G["nearfar_aabb"])
    N = ro.shape[0]
    for mn in (0.2, 0.05):
        nears, fars = torch.empty(N), torch.empty(N)
        oracle.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, N, mn, nears, fars)
        want = G[f"nearfar_min{mn:g}"]
        got = np.stack([nears.numpy(), fars.numpy()], -1)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.argwhere(got.view(np.uint32) != want.view(np.uint32))[:5]
    miss = want[:, 0] == np.finfo(np.float32).max
    assert 0 < miss.sum() < N
