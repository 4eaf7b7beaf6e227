"""Second witness for the two oracle files whose arithmetic no reference fixture pins (TEST INFRASTRUCTURE ONLY).

`oracle/src/s3o_gridencoder.c` and `s3o_raymarching.c` restate the reference's CUDA kernels statement by statement; the
reference ships no golden outputs for them and its CUDA cannot run here, so a transcription error in the restatement would
go unnoticed (the HIP kernels were written from the same reading).  This module states the same MATHEMATICS a second time,
independently of the C text: vectorised numpy, float64, from the definitions —

  * multiresolution grid encoding (Instant-NGP §3; gridencoder.cu:87-242 only for the conventions that are not in the
    paper: per-level scale `2^(l*S)*H - 1`, resolution `ceil(scale)+1`, the `+0.5` cell offset without align_corners,
    dense index while the running stride fits the level's table, else `xor_d (x_d * prime_d)`, all in uint32, `% size`);
  * emission-absorption compositing with early termination (NeRF eq. 3; raymarching.cu:501-693 for the termination rule
    `T < T_thresh` checked after the sample is accumulated, and the depth convention `t = deltas[:,1]` cumulative),
    and its analytic gradient obtained here by DIFFERENTIATING THE FORWARD NUMERICALLY-EXACTLY in float64 (closed form of
    d/dsigma of the discrete sum), not by copying the kernel's recurrence;
  * morton codes by bit interleaving, packbits by numpy.packbits(bitorder='little').

Agreement of the C oracle (fp32) with this witness (float64) within fp32 rounding, and exact agreement of every integer
(cell coordinates, table rows, morton codes, packed bits), is checked in tests/test_witness.py."""
import numpy as np

PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], dtype=np.uint64)


def level_geometry(L, S, H):
    """per-level (scale, resolution) in the reference's float32 arithmetic (the only place fp32 matters: it fixes cells)"""
    lv = np.arange(L, dtype=np.float32)
    # exp2f(level * S) rounded to float32, then `* H - 1` with ONE rounding (nvcc contracts a * b - c into an fma)
    e = np.exp2((lv * np.float32(S)).astype(np.float64)).astype(np.float32)
    scale = (e.astype(np.float64) * float(H) - 1.0).astype(np.float32)
    return scale, np.ceil(scale).astype(np.uint32) + 1


def grid_rows(cells, size, resolution, gridtype="hash", align_corners=False):
    """table row of integer cell coordinates `cells` [..., D] (uint32 wrap-around arithmetic)"""
    D = cells.shape[-1]
    cells = cells.astype(np.uint64)
    stride, index = np.uint64(1), np.zeros(cells.shape[:-1], dtype=np.uint64)
    dense = True
    for d in range(D):
        if stride <= size:
            index = (index + cells[..., d] * stride) & np.uint64(0xFFFFFFFF)
            stride = (stride * np.uint64(resolution if align_corners else resolution + 1)) & np.uint64(0xFFFFFFFF)
    if gridtype == "hash" and stride > size:
        index = np.zeros(cells.shape[:-1], dtype=np.uint64)
        for d in range(D):
            index ^= (cells[..., d] * PRIMES[d]) & np.uint64(0xFFFFFFFF)
        dense = False
    return (index % np.uint64(size)).astype(np.int64), dense


def grid_encode(x01, table, offsets, S, H, gridtype="hash", align_corners=False, smoothstep=False):
    """x01 [B, D] in [0, 1] (float32 values), table [rows, C] -> features [B, L*C] float64, rows [B, L, 2^D], weights"""
    B, D = x01.shape
    L = len(offsets) - 1
    C = table.shape[1]
    scale, res = level_geometry(L, S, H)
    out = np.zeros((B, L, C))
    all_rows = np.zeros((B, L, 1 << D), dtype=np.int64)
    all_w = np.zeros((B, L, 1 << D))
    x32 = x01.astype(np.float32)
    inside = ((x32 >= 0) & (x32 <= 1)).all(1)
    for l in range(L):
        size = int(offsets[l + 1] - offsets[l])
        # the cell is decided in float32 exactly as the kernels do (fused multiply-add, then floor); the fractional
        # position and everything after it is float64
        pos32 = (x32.astype(np.float64) * float(scale[l]) + (0.0 if align_corners else 0.5)).astype(np.float32)
        cell = np.floor(pos32).astype(np.int64)
        frac = pos32.astype(np.float64) - cell
        if smoothstep:
            frac = frac * frac * (3.0 - 2.0 * frac)
        for corner in range(1 << D):
            bits = np.array([(corner >> d) & 1 for d in range(D)])
            w = np.prod(np.where(bits, frac, 1.0 - frac), axis=1)
            rows, _ = grid_rows((cell + bits).astype(np.uint64), size, int(res[l]), gridtype, align_corners)
            all_rows[:, l, corner], all_w[:, l, corner] = rows, w
            out[:, l] += w[:, None] * table[int(offsets[l]) + rows].astype(np.float64)
    out[~inside] = 0
    all_w[~inside] = 0
    return out.reshape(B, L * C), all_rows, all_w


def grid_encode_backward(grad, rows, weights, offsets, n_rows, C):
    """grad [B, L*C] -> table gradient [n_rows, C] (float64 scatter-add of w * grad)"""
    B, L, K = rows.shape
    g = np.zeros((n_rows, C))
    grad = grad.reshape(B, L, C).astype(np.float64)
    for l in range(L):
        for k in range(K):
            np.add.at(g, int(offsets[l]) + rows[:, l, k], weights[:, l, k][:, None] * grad[:, l])
    return g


def composite_train(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    """rays [N, 3] (index, offset, count) -> weights_sum [N], depth [N], image [N, 3]; per-sample weights and the kept count"""
    N = rays.shape[0]
    ws, depth, image = np.zeros(N), np.zeros(N), np.zeros((N, 3))
    weights = np.zeros(sigmas.shape[0])
    kept = np.zeros(N, dtype=np.int64)
    for n in range(N):
        idx, off, cnt = (int(v) for v in rays[n])
        s = sigmas[off:off + cnt].astype(np.float64)
        dl = deltas[off:off + cnt].astype(np.float64)
        alpha = 1.0 - np.exp(-s * dl[:, 0])
        T = np.concatenate([[1.0], np.cumprod(1.0 - alpha)])  # transmittance BEFORE each sample, then after the last
        # a sample is accumulated, THEN the ray stops once the remaining transmittance is below the threshold
        stop = np.nonzero(T[1:] < T_thresh)[0]
        k = cnt if stop.size == 0 else int(stop[0]) + 1
        w = alpha[:k] * T[:k]
        weights[off:off + k] = w
        kept[n] = k
        t = np.cumsum(dl[:k, 1])
        ws[idx], depth[idx] = w.sum(), (w * t).sum()
        image[idx] = (w[:, None] * rgbs[off:off + k].astype(np.float64)).sum(0)
    return ws, depth, image, weights, kept


def composite_train_grads(g_ws, g_image, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    """d(loss)/d sigma, d(loss)/d rgb with loss = <g_ws, weights_sum> + <g_image, image>, differentiating
    w_i = (1 - exp(-s_i d_i)) * prod_{j<i} exp(-s_j d_j) directly:
      dw_i/ds_i = d_i * (T_i - w_i) = d_i * T_{i+1},   dw_k/ds_i = -d_i * w_k  (k > i)"""
    ws, depth, image, weights, kept = composite_train(sigmas, rgbs, deltas, rays, T_thresh)
    gs, gc = np.zeros(sigmas.shape[0]), np.zeros((sigmas.shape[0], 3))
    for n in range(rays.shape[0]):
        idx, off, cnt = (int(v) for v in rays[n])
        k = int(kept[n])
        w = weights[off:off + k]
        c = rgbs[off:off + k].astype(np.float64)
        d = deltas[off:off + k, 0].astype(np.float64)
        s = sigmas[off:off + k].astype(np.float64)
        T_after = np.cumprod(np.exp(-s * d))
        per = g_ws[idx] + c @ g_image[idx].astype(np.float64)      # d loss / d w_k
        tail = np.concatenate([np.cumsum((w * per)[::-1])[::-1][1:], [0.0]])  # sum_{k>i} w_k * per_k
        gs[off:off + k] = d * (T_after * per - tail)
        gc[off:off + k] = w[:, None] * g_image[idx].astype(np.float64)
    return gs, gc


def morton3d(xyz):
    """bit interleave x (bit 0), y (bit 1), z (bit 2) of 10-bit coordinates"""
    code = np.zeros(xyz.shape[0], dtype=np.uint64)
    for b in range(10):
        for d in range(3):
            code |= ((xyz[:, d].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + d)
    return code.astype(np.int64)


def packbits(grid, thresh):
    return np.packbits((grid.reshape(-1) > thresh).astype(np.uint8), bitorder="little")


# ---------------------------------------------------------------------------------------------------------------------
# Ray marching: near/far slab test and the two-pass DDA of march_rays_train, stated a second time.
#
# Written from raymarching.cu:92-156 (near_far_from_aabb), :20-60 (clamp / signf / mip_from_pos / mip_from_dt / morton) and
# :312-480 (kernel_march_rays_train) WITHOUT consulting oracle/src/s3o_raymarching.c: one Python loop per ray, every
# intermediate an explicit numpy float32 (or the float64 the CUDA expression promotes to: `0.5 * (...) * H` has a double
# literal), nvcc's default contraction of `a * b + c` into one fused multiply-add spelled out (fma32 below).  Ray-ordered
# span reservation (the reference's atomicAdd order is arbitrary; ray order is the deterministic one the build pins).
# tests/test_witness.py compares per-ray sample counts, span offsets, counters and every sample position / delta with the C
# oracle bit for bit.
F32 = np.float32


def fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in 80-bit extended precision, one rounding to float32"""
    return F32(np.longdouble(a) * np.longdouble(b) + np.longdouble(c))


def _clamp32(x, lo, hi):
    return F32(min(F32(hi), max(F32(lo), F32(x))))  # fminf(max, fmaxf(min, x))


def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
    """raymarching.cu:92-147: slab test on x, then y, then z; a miss -> both FLT_MAX; near clamped from below by min_near"""
    N = rays_o.shape[0]
    nears, fars = np.empty(N, dtype=F32), np.empty(N, dtype=F32)
    fmax = np.finfo(F32).max
    a = aabb.astype(F32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for n in range(N):
            o, d = rays_o[n].astype(F32), rays_d[n].astype(F32)
            rd = F32(1) / d
            near, far = (a[0] - o[0]) * rd[0], (a[3] - o[0]) * rd[0]
            if near > far:
                near, far = far, near
            miss = False
            for ax in (1, 2):
                n2, f2 = (a[ax] - o[ax]) * rd[ax], (a[3 + ax] - o[ax]) * rd[ax]
                if n2 > f2:
                    n2, f2 = f2, n2
                if near > f2 or n2 > far:
                    miss = True
                    break
                if n2 > near:
                    near = n2
                if f2 < far:
                    far = f2
            if miss:
                nears[n] = fars[n] = fmax
                continue
            if near < F32(min_near):
                near = F32(min_near)
            nears[n], fars[n] = near, far
    return nears, fars


def _expand_bits(v):
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v & 0xFFFFFFFF


def _morton(x, y, z):
    return _expand_bits(x) | (_expand_bits(y) << 1) | (_expand_bits(z) << 2)


def _frexp_exponent(v):
    return int(np.frexp(F32(v))[1])  # frexpf(0) -> exponent 0


def _mip_from_pos(x, y, z, C):
    e = _frexp_exponent(max(abs(F32(x)), abs(F32(y)), abs(F32(z))))
    return int(min(C - 1, max(0, e)))


def _mip_from_dt(dt, H, C):
    e = _frexp_exponent(F32(np.float64(F32(dt) * F32(H)) * 0.5))  # `dt * H * 0.5`: float product, double literal
    return int(min(C - 1, max(0, e)))


def _march_kit(bitfield, bound, dt_gamma, max_steps, C, H):
    """the per-sample pieces the training and the inference marcher share: cell(t, o, d) — position, step, mip level,
    occupancy of the sample at t — and skip(...) — the t sequence advanced behind the current voxel"""
    bound, dt_gamma = F32(bound), F32(dt_gamma)
    rH = F32(1) / F32(H)
    H3 = H * H * H
    SQRT3 = F32(1.7320508075688772)
    dt_min = F32(2) * SQRT3 / F32(max_steps)
    dt_max = F32(2) * SQRT3 * F32(1 << (C - 1)) / F32(H)
    bits = np.asarray(bitfield, dtype=np.uint8)

    def cell(t, o, d):
        x = _clamp32(fma32(t, d[0], o[0]), -bound, bound)  # ox + t * dx, contracted
        y = _clamp32(fma32(t, d[1], o[1]), -bound, bound)
        z = _clamp32(fma32(t, d[2], o[2]), -bound, bound)
        dt = _clamp32(t * dt_gamma, dt_min, dt_max)
        level = max(_mip_from_pos(x, y, z, C), _mip_from_dt(dt, H, C))
        mip_bound = F32(min(F32(np.ldexp(F32(1), level)), bound))
        mip_rbound = F32(1) / mip_bound

        def grid_coord(v):  # clamp(0.5 * (v * mip_rbound + 1) * H, 0.0f, (float)(H - 1)) -> int (truncation)
            inner = fma32(v, mip_rbound, F32(1))            # float: v * mip_rbound + 1, contracted
            val = np.float64(0.5) * np.float64(inner) * np.float64(H)  # double: 0.5 is a double literal, H converts exactly
            return int(_clamp32(F32(val), F32(0), F32(H - 1)))
        nx, ny, nz = grid_coord(x), grid_coord(y), grid_coord(z)
        index = level * H3 + _morton(nx, ny, nz)
        occ = (int(bits[index // 8]) >> (index % 8)) & 1
        return x, y, z, dt, mip_bound, (nx, ny, nz), bool(occ)

    def skip(t, x, y, z, n3, mip_bound, d, rd):
        """distance to the next voxel of this mip level, then steps of the t sequence until it is passed"""
        def axis(nc, dc, c, rdc):
            sgn = F32(np.copysign(F32(1), dc))
            edge = F32(nc) + F32(0.5) + F32(0.5) * sgn              # half-integers: exact
            u = F32(F32(edge * rH) * F32(2)) - F32(1)                  # (edge * rH * 2 - 1): the doubling is exact
            return F32(F32(F32(u * mip_bound) - c) * rdc)             # mip_bound is a power of two (or the bound itself)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            tx, ty, tz = axis(n3[0], d[0], x, rd[0]), axis(n3[1], d[1], y, rd[1]), axis(n3[2], d[2], z, rd[2])
            tt = F32(t + F32(max(F32(0), min(tx, min(ty, tz)))))
        while True:
            t = F32(t + _clamp32(t * dt_gamma, dt_min, dt_max))
            if not (t < tt):
                return t

    return cell, skip, dt_min, dt_max, dt_gamma


def march_rays_train(rays_o, rays_d, bitfield, bound, dt_gamma, max_steps, C, H, M, nears, fars, noises):
    """raymarching.cu:312-480 -> (xyzs [M,3], dirs [M,3], deltas [M,2] float32 zero-initialised, rays [N,3] int32
    (id, offset, count), counter [2])"""
    N = rays_o.shape[0]
    cell, skip, dt_min, dt_max, dt_gamma = _march_kit(bitfield, bound, dt_gamma, max_steps, C, H)
    xyzs, dirs, deltas = np.zeros((M, 3), F32), np.zeros((M, 3), F32), np.zeros((M, 2), F32)
    rays = np.zeros((N, 3), np.int32)
    point_index = 0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for n in range(N):
            o, d = rays_o[n].astype(F32), rays_d[n].astype(F32)
            rd = F32(1) / d
            near, far, noise = F32(nears[n]), F32(fars[n]), F32(noises[n])
            t0 = fma32(_clamp32(near * dt_gamma, dt_min, dt_max), noise, near)  # t0 += clamp(...) * noise, contracted
            # first pass: count
            t, num = t0, 0
            while t < far and num < max_steps:
                x, y, z, dt, mip_bound, n3, occ = cell(t, o, d)
                if occ:
                    num += 1
                    t = F32(t + dt)
                else:
                    t = skip(t, x, y, z, n3, mip_bound, d, rd)
            rays[n] = (n, point_index, num)
            start = point_index
            point_index += num
            if num == 0 or start + num > M:
                continue
            # second pass: write
            t, step, last_t = t0, 0, t0
            while t < far and step < num:
                x, y, z, dt, mip_bound, n3, occ = cell(t, o, d)
                if occ:
                    xyzs[start + step] = (x, y, z)
                    dirs[start + step] = d
                    t = F32(t + dt)
                    deltas[start + step] = (dt, F32(t - last_t))
                    last_t = t
                    step += 1
                else:
                    t = skip(t, x, y, z, n3, mip_bound, d, rd)
    return xyzs, dirs, deltas, rays, np.array([point_index, N], dtype=np.int32)


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bitfield, bound, dt_gamma, max_steps, C, H, fars, noises):
    """One iteration of the inference loop (raymarching.cu:701-800): every alive ray continues from its own t and records up
    to n_step samples; slots it does not fill stay zero.  -> xyzs [n_alive * n_step, 3], dirs, deltas [.., 2] float32"""
    cell, skip, dt_min, dt_max, dt_gamma = _march_kit(bitfield, bound, dt_gamma, max_steps, C, H)
    M = n_alive * n_step
    xyzs, dirs, deltas = np.zeros((M, 3), F32), np.zeros((M, 3), F32), np.zeros((M, 2), F32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for n in range(n_alive):
            ray = int(rays_alive[n])
            o, d = rays_o[ray].astype(F32), rays_d[ray].astype(F32)
            rd = F32(1) / d
            far = F32(fars[ray])
            t = F32(rays_t[ray])
            t = fma32(_clamp32(t * dt_gamma, dt_min, dt_max), F32(noises[n]), t)  # t += clamp(...) * noise, contracted
            last_t, step, row = t, 0, n * n_step
            while t < far and step < n_step:
                x, y, z, dt, mip_bound, n3, occ = cell(t, o, d)
                if occ:
                    xyzs[row + step] = (x, y, z)
                    dirs[row + step] = d
                    t = F32(t + dt)
                    deltas[row + step] = (dt, F32(t - last_t))
                    last_t = t
                    step += 1
                else:
                    t = skip(t, x, y, z, n3, mip_bound, d, rd)
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """The compositing half of an iteration (raymarching.cu:808-895), in float64: w = alpha * (1 - sum of earlier w), a chunk
    ends at the first empty slot (delta == 0) or once the transmittance in front of a sample has fallen below T_thresh — that
    sample still counts; a ray whose chunk ended early is dead (-1), the others advance their t.  In place on copies;
    -> (rays_alive, rays_t, weights_sum, depth, image)"""
    alive, rt = np.array(rays_alive, np.int32), np.array(rays_t, np.float64)
    ws, dp, im = np.array(weights_sum, np.float64), np.array(depth, np.float64), np.array(image, np.float64)
    for n in range(n_alive):
        ray = int(alive[n])
        t, step = rt[ray], 0
        while step < n_step:
            k = n * n_step + step
            if deltas[k, 0] == 0:
                break
            alpha = 1.0 - np.exp(-np.float64(sigmas[k]) * np.float64(deltas[k, 0]))
            T = 1.0 - ws[ray]
            w = alpha * T
            ws[ray] += w
            t += np.float64(deltas[k, 1])
            dp[ray] += w * t
            im[ray] += w * np.asarray(rgbs[k], np.float64)
            if T < T_thresh:
                break
            step += 1
        if step < n_step:
            alive[n] = -1
        else:
            rt[ray] = t
    return alive, rt, ws, dp, im
