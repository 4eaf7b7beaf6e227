"""CPU restatement (numpy) of TensoRF's vector-matrix features — TEST INFRASTRUCTURE ONLY (tests/ may import it; the
product path never does).

Follows tensoRF/network.py:112-153 of the reference (`get_sigma_feat`, `get_color_feat`): per component i, plane factor
[R, H, W] sampled at (x[mat_ids[i][0]] -> W, x[mat_ids[i][1]] -> H) and line factor [R, D] sampled at x[vec_ids[i]], both
with `F.grid_sample(..., align_corners=True)` (bilinear, zeros padding; the line through a width-1 image), multiplied and
(density) summed.  The sampling itself lives in a third-party dependency of the reference — PyTorch (this image: 2.10),
aten/src/ATen/native/GridSampler.h / cuda/GridSampler.cu — whose published algorithm is restated here:
    index = ((c + 1) / 2) * (size - 1);  x0 = floor(ix), y0 = floor(iy)
    nw = (x0+1 - ix)(y0+1 - iy), ne = (ix - x0)(y0+1 - iy), sw = (x0+1 - ix)(iy - y0), se = (ix - x0)(iy - y0)
    out = sum over the IN-RANGE corners of value * weight, in the order nw, ne, sw, se
and the backward: every in-range corner receives grad_out * weight (scatter-add).
Pinned by tests/test_vm_oracle.py against torch's own CPU grid_sample on the reference's call sequence.
"""
import numpy as np

MAT_IDS = ((0, 1), (0, 2), (1, 2))  # tensoRF/network.py:37
VEC_IDS = (2, 1, 0)                 # tensoRF/network.py:38


def _locate(c, size):
    idx = ((c.astype(np.float32) + np.float32(1)) / np.float32(2)) * np.float32(size - 1)
    i0 = np.floor(idx)
    w1 = (idx - i0).astype(np.float32)
    w0 = ((i0 + np.float32(1)) - idx).astype(np.float32)
    return i0.astype(np.int64), w0, w1


def _sample_plane(P, x0, y0, wx0, wx1, wy0, wy1):
    """P [R,H,W] -> [R,N]; corner order nw, ne, sw, se"""
    R, H, W = P.shape
    out = np.zeros((R, x0.shape[0]), np.float32)
    for dx, dy, w in ((0, 0, wx0 * wy0), (1, 0, wx1 * wy0), (0, 1, wx0 * wy1), (1, 1, wx1 * wy1)):
        xx, yy = x0 + dx, y0 + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = P[:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        out += np.where(ok, v * w.astype(np.float32), np.float32(0)).astype(np.float32)
    return out


def _sample_line(L, z0, wz0, wz1):
    R, D = L.shape
    out = np.zeros((R, z0.shape[0]), np.float32)
    for dz, w in ((0, wz0), (1, wz1)):
        zz = z0 + dz
        ok = (zz >= 0) & (zz < D)
        out += np.where(ok, L[:, np.clip(zz, 0, D - 1)] * w, np.float32(0)).astype(np.float32)
    return out


def vm_products(x, planes, lines):
    """x [N,3] in [-1,1]; planes[i] [R_i,H,W]; lines[i] [R_i,D] -> list of [R_i,N] products plane*line"""
    out = []
    for i in range(3):
        P, L = planes[i], lines[i]
        x0, wx0, wx1 = _locate(x[:, MAT_IDS[i][0]], P.shape[2])
        y0, wy0, wy1 = _locate(x[:, MAT_IDS[i][1]], P.shape[1])
        z0, wz0, wz1 = _locate(x[:, VEC_IDS[i]], L.shape[1])
        out.append(_sample_plane(P, x0, y0, wx0, wx1, wy0, wy1) * _sample_line(L, z0, wz0, wz1))
    return out


def sigma_feat(x, planes, lines):
    """tensoRF/network.py:112-130"""
    total = np.zeros(x.shape[0], np.float32)
    for p in vm_products(x, planes, lines):
        total = total + p.sum(0, dtype=np.float32)
    return total


def color_products(x, planes, lines):
    """tensoRF/network.py:133-147 before `.T` / basis_mat: [sum R_i, N]"""
    return np.concatenate(vm_products(x, planes, lines), 0)


def factor_grads(x, planes, lines, grad_rows):
    """grad_rows [sum R_i, N] (gradient of the products; for the density: the same [N] row repeated) -> (d planes, d lines)"""
    gp, gl, row = [], [], 0
    for i in range(3):
        P, L = planes[i], lines[i]
        R, H, W = P.shape
        D = L.shape[1]
        g = grad_rows[row:row + R].astype(np.float64)
        row += R
        x0, wx0, wx1 = _locate(x[:, MAT_IDS[i][0]], W)
        y0, wy0, wy1 = _locate(x[:, MAT_IDS[i][1]], H)
        z0, wz0, wz1 = _locate(x[:, VEC_IDS[i]], D)
        m = _sample_plane(P, x0, y0, wx0, wx1, wy0, wy1).astype(np.float64)
        l = _sample_line(L, z0, wz0, wz1).astype(np.float64)
        dP, dL = np.zeros(P.shape, np.float64), np.zeros(L.shape, np.float64)
        r_idx = np.arange(R)[:, None]
        for dx, dy, w in ((0, 0, wx0 * wy0), (1, 0, wx1 * wy0), (0, 1, wx0 * wy1), (1, 1, wx1 * wy1)):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            np.add.at(dP, (r_idx, np.clip(yy, 0, H - 1)[None], np.clip(xx, 0, W - 1)[None]), np.where(ok, g * l * w, 0.0))
        for dz, w in ((0, wz0), (1, wz1)):
            zz = z0 + dz
            ok = (zz >= 0) & (zz < D)
            np.add.at(dL, (r_idx, np.clip(zz, 0, D - 1)[None]), np.where(ok, g * m * w, 0.0))
        gp.append(dP.astype(np.float32))
        gl.append(dL.astype(np.float32))
    return gp, gl
