"""Host logic of the planned hash-grid forward (csrc/gridencoder.hip: make_forward_plan) — no GPU needed: the LDS jobs cover
their levels' rows and points exactly once, the gather segments cover every (level, chunk) exactly once, every XCD gets the
same share, and at most two large tables land in one XCD's L2."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))


def _offsets(L=16, base=16, log2T=19, desired=2048, D=3):
    pls = np.exp2(np.log2(desired / base) / (L - 1))
    offs, off = [], 0
    for i in range(L):
        res = int(np.ceil(base * pls ** i))
        n = int(np.ceil(min(2 ** log2T, (res + 1) ** D) / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return offs, float(np.log2(pls))


def _describe(lib, B, Cc, dtype, offs, S, L=16, D=3, H=16):
    arr = (C.c_int32 * len(offs))(*offs)
    out = (C.c_uint32 * 4096)()
    lib.s3d_grid_forward_plan_describe.restype = C.c_int
    n = lib.s3d_grid_forward_plan_describe(C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                           C.c_int(dtype), C.c_uint32(0), C.c_int(0), arr, out, C.c_size_t(4096))
    if n <= 0:
        return None
    w = list(out[:n])
    n_jobs, n_job_wg, chunk = w[:3]
    jobs = [tuple(w[3 + 6 * j: 9 + 6 * j]) for j in range(n_jobs)]
    p = 3 + 6 * n_jobs
    xcds = []
    for _ in range(8):
        nseg, wg = w[p], w[p + 1]
        segs = [tuple(w[p + 2 + 3 * i: p + 5 + 3 * i]) for i in range(nseg)]
        p += 2 + 3 * nseg
        xcds.append((wg, segs))
    assert p == n
    return n_job_wg, chunk, jobs, xcds


@pytest.fixture(scope="module")
def lib():
    import s3d_hip
    s3d_hip.build()
    return C.CDLL(s3d_hip.LIB_PATH)


@pytest.mark.parametrize("B,Cc,dtype", [(262144, 2, 1), (279248, 2, 1), (1 << 21, 2, 1), (262144, 2, 0), (100000, 4, 1)])
def test_plan_covers_everything_once(lib, B, Cc, dtype):
    offs, S = _offsets()
    d = _describe(lib, B, Cc, dtype, offs, S)
    assert d is not None
    n_job_wg, chunk, jobs, xcds = d
    L = 16
    elem = 2 if dtype == 1 else 4
    NC = -(-B // chunk)
    job_levels = sorted({j[0] for j in jobs})
    assert n_job_wg % 8 == 0 and n_job_wg >= sum(j[5] for j in jobs)
    for l in job_levels:
        js = sorted([j for j in jobs if j[0] == l], key=lambda j: j[3])
        rows = offs[l + 1] - offs[l]
        assert js[0][3] == 0 and js[-1][4] == 0xFFFFFFFF
        for a, b in zip(js, js[1:]):
            assert a[4] == b[3], "selection ranges of a split level are contiguous"
        res = int(np.ceil(np.float32(np.exp2(np.float32(l * S)) * 16 - 1))) + 1
        span = 1 + (res + 1) + (res + 1) ** 2
        for (_, row_lo, nrows, sel_lo, sel_hi, groups) in js:
            assert row_lo % 8 == 0 and nrows * Cc * elem <= 75 * 1024 and row_lo + nrows <= rows and 1 <= groups <= NC
            last = min(sel_hi, rows) - 1          # largest corner-0 row served by the job
            if len(js) > 1:
                assert row_lo <= sel_lo and last + span <= row_lo + nrows + 0 or row_lo + nrows == rows, "staged rows cover every corner of the selected points"
    # gather levels: every (level, chunk) exactly once
    seen = {}
    for wg, segs in xcds:
        assert wg == sum(s[2] for s in segs)
        for (l, c0, n) in segs:
            assert l not in job_levels
            cov = seen.setdefault(l, np.zeros(NC, dtype=np.int32))
            cov[c0:c0 + n] += 1
    for l in range(L):
        if l in job_levels:
            assert l not in seen
        else:
            assert l in seen and (seen[l] == 1).all(), f"level {l}: chunks not covered exactly once"
    wgs = [wg for wg, _ in xcds]
    assert max(wgs) - min(wgs) <= 2 + len(seen), f"unbalanced XCDs: {wgs}"
    big = {l for l in seen if (offs[l + 1] - offs[l]) * Cc * elem > (1 << 20)}
    for wg, segs in xcds:
        assert len({s[0] for s in segs} & big) <= 3


def test_plan_lego_fp16_shape(lib):
    """The Lego configuration under -O: levels 0-4 are dense -> LDS jobs (0 and 1 whole, 2-4 split by rows); the eleven hashed
    levels are gathered, 1.375 levels' worth of chunks per XCD."""
    offs, S = _offsets()
    n_job_wg, chunk, jobs, xcds = _describe(lib, 262144, 2, 1, offs, S)
    levels = sorted({j[0] for j in jobs})
    assert levels[:2] == [0, 1] and 2 in levels
    whole = [j for j in jobs if j[0] in (0, 1)]
    assert all(j[1] == 0 and j[3] == 0 for j in whole)
    gathered = {s[0] for _, segs in xcds for s in segs}
    assert gathered == set(range(16)) - set(levels) and set(range(5, 16)) <= gathered


def test_small_batches_take_the_plain_launch(lib):
    offs, S = _offsets()
    assert _describe(lib, 4096, 2, 1, offs, S) is None
    assert _describe(lib, 65535, 2, 1, offs, S) is None
