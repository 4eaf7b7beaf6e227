"""GPU parity: libseal3d_hip raymarching entry points vs the CPU oracle on the same seeded inputs.
Integer / index / compaction results must be BIT-EXACT; compositing is FP (rtol 1e-4, north_star)."""
import numpy as np
import pytest
import torch

from nerf import synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(seed=0, cascade=1, bound=1.0):
    grid, bits = syn.lego_like_density_grid(seed=seed, cascade=cascade, bound=bound)
    return torch.from_numpy(grid), torch.from_numpy(bits)


def _rays(n, seed=0, H=800, W=800):
    poses = syn.orbit_poses(4, seed=seed)
    g = torch.Generator().manual_seed(seed)
    r = syn.get_rays(poses[seed % 4:seed % 4 + 1], syn.lego_intrinsics(H, W), H, W, N=n, generator=g)
    return r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()


def _both(oracle, hip, name, cpu_args, n_out_tensors):
    """run backend fn `name` on the oracle (CPU tensors) and on HIP (cuda copies); return both arg lists"""
    gpu_args = [a.cuda() if torch.is_tensor(a) else a for a in cpu_args]
    getattr(oracle.RaymarchingBackend, name)(*cpu_args)
    getattr(hip.RaymarchingBackend, name)(*gpu_args)
    torch.cuda.synchronize()
    return cpu_args, gpu_args


def test_morton_roundtrip_full_grid(oracle, hip):
    idx = torch.arange(128)
    c = torch.stack(torch.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3).int().contiguous()
    N = c.shape[0]
    out_c = torch.empty(N, dtype=torch.int32)
    cpu, gpu = _both(oracle, hip, "morton3D", [c, N, out_c], 1)
    assert torch.equal(cpu[2], gpu[2].cpu())
    # size-independent property: invert(morton(c)) == c, indices are a permutation of [0, 128^3)
    back = torch.empty(N, 3, dtype=torch.int32, device="cuda")
    hip.RaymarchingBackend.morton3D_invert(gpu[2], N, back)
    assert torch.equal(back.cpu(), c)
    assert torch.equal(torch.sort(gpu[2].cpu().long()).values, torch.arange(N))


def test_morton_invert_random_incl_out_of_range(oracle, hip):
    g = torch.Generator().manual_seed(3)
    ind = torch.randint(0, 2 ** 30, (10000,), generator=g, dtype=torch.int64).int()
    out = torch.empty(10000, 3, dtype=torch.int32)
    cpu, gpu = _both(oracle, hip, "morton3D_invert", [ind, 10000, out], 1)
    assert torch.equal(cpu[2], gpu[2].cpu())


def test_packbits(oracle, hip):
    g = torch.Generator().manual_seed(5)
    grid = torch.rand(2, 128 ** 3, generator=g) * 2 - 0.5
    grid[0, :64] = -1.0
    N = grid.numel() // 8
    bf = torch.empty(N, dtype=torch.uint8)
    cpu, gpu = _both(oracle, hip, "packbits", [grid, N, 0.37, bf], 1)
    assert torch.equal(cpu[3], gpu[3].cpu())
    # known answer: numpy packbits little-endian
    ref = np.packbits((grid.numpy().reshape(-1) > np.float32(0.37)).astype(np.uint8), bitorder="little")
    assert np.array_equal(ref, gpu[3].cpu().numpy())


def test_near_far_incl_misses_and_axis_parallel(oracle, hip):
    ro, rd = _rays(4096, seed=1)
    ro, rd = ro.clone(), rd.clone()
    rd[:8] = torch.tensor([0.0, 0.0, 1.0])        # axis-parallel: 1/0 = inf in two slabs
    ro[8:16] = torch.tensor([5.0, 5.0, 5.0])      # misses
    rd[8:16] = torch.tensor([0.0, 1.0, 0.0])
    ro[16:24] = torch.tensor([0.0, 0.0, 0.0])     # origin inside the box: near clamps to min_near
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    N = ro.shape[0]
    nears, fars = torch.empty(N), torch.empty(N)
    cpu, gpu = _both(oracle, hip, "near_far_from_aabb", [ro, rd, aabb, N, 0.2, nears, fars], 2)
    a, b = cpu[5].numpy(), gpu[5].cpu().numpy()
    # bit-exact incl. NaN patterns from inf*0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(cpu[6].numpy().view(np.uint32), gpu[6].cpu().numpy().view(np.uint32))
    assert (cpu[5][16:24] == 0.2).all()


def test_near_far_draws_the_per_ray_jitter_from_a_device_step_number(oracle, hip):
    """build extension of near_far_from_aabb: nears/fars unchanged, noises = u01(key, *step, ray) in [0, 1) — uniform, a new
    draw for every step number and key, reproducible for equal ones; march_rays_train(noises=...) uses it as given"""
    R = hip.RaymarchingBackend
    ro, rd = _rays(1 << 16, seed=5)
    ro, rd = ro.cuda(), rd.cuda()
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device="cuda")
    N = ro.shape[0]
    ref = [torch.empty(N, device="cuda") for _ in range(2)]
    R.near_far_from_aabb(ro, rd, aabb, N, 0.2, *ref)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    draws = {}
    for key in (1, 2):
        for st in (0, 1, 7):
            step.fill_(st)
            out = [torch.empty(N, device="cuda") for _ in range(3)]
            R.near_far_from_aabb(ro, rd, aabb, N, 0.2, out[0], out[1], noises=out[2], noise_step=step, noise_key=key)
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
            u = out[2]
            assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
            assert abs(float(u.mean()) - 0.5) < 0.01 and abs(float(u.var()) - 1 / 12) < 0.005
            hist = torch.histc(u, bins=16, min=0, max=1) / N
            assert float((hist - 1 / 16).abs().max()) < 0.01
            assert abs(float(torch.corrcoef(torch.stack([u[:-1], u[1:]]))[0, 1])) < 0.02  # neighbouring rays independent
            draws[(key, st)] = u
    keys = list(draws)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            assert abs(float(torch.corrcoef(torch.stack([draws[keys[i]], draws[keys[j]]]))[0, 1])) < 0.02
    step.fill_(7)
    again = torch.empty(N, device="cuda")
    R.near_far_from_aabb(ro, rd, aabb, N, 0.2, torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), noises=again,
                         noise_step=step, noise_key=2)
    assert torch.equal(again, draws[(2, 7)])


def test_native_occupancy_sweep_against_the_torch_rule(hip):
    """s3d_sweep_draw / s3d_sweep_update (the graph-replayed update_extra_state steady state) against the torch op sequence of
    nerf/renderer.py: same cells for the same uniforms, positions inside their cells, EMA-max update bit for bit (the largest
    sample of a cell where several fall into it), never-seen cells (-1) and NaN samples left alone, fixed-order sum."""
    import s3d_hip
    R = hip.RaymarchingBackend
    H, N = 64, 64 ** 3 // 4
    H3 = H ** 3
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    grid = torch.rand(H3, device=dev, generator=g)
    grid[grid < 0.8] = 0
    grid[:777] = -1
    u1 = torch.sort(torch.rand(N, device=dev, dtype=torch.float64, generator=g)).values
    u2 = torch.sort(torch.rand(N, device=dev, dtype=torch.float64, generator=g)).values
    csum = torch.cumsum(grid > 0, dim=0, dtype=torch.int32)
    step = torch.full((1,), 5, dtype=torch.int32, device=dev)
    bound, hgs = 1.0, 1.0 / H
    cells, xyzs = R.sweep_draw(u1, u2, csum, H, bound, hgs, 99, step)
    ref_uniform = (u1 * H3).long().clamp_(max=H3 - 1)
    ref_occ = torch.searchsorted(csum, (u2 * csum[-1]).to(torch.int32), right=True).clamp_(max=H3 - 1)
    assert torch.equal(cells.long(), torch.cat([ref_uniform, ref_occ]))
    assert bool((grid[cells[N:].long()] > 0).all())
    coords = torch.empty(2 * N, 3, dtype=torch.int32, device=dev)
    R.morton3D_invert(cells, 2 * N, coords)
    centre = (2 * coords.float() / (H - 1) - 1) * (bound - hgs)
    off = (xyzs - centre) / hgs
    assert float(off.abs().max()) <= 1.0 + 1e-5 and abs(float(off.mean())) < 5e-3 and abs(float(off.var()) - 1 / 3) < 5e-3
    cells2, xyzs2 = R.sweep_draw(u1, u2, csum, H, bound, hgs, 99, step + 1)
    assert torch.equal(cells, cells2) and not torch.equal(xyzs, xyzs2)  # fresh jitter for the next step number
    # update: samples incl. duplicates, exact zeros, a NaN
    for dt in (torch.float32, torch.float16):
        sigma = (torch.rand(2 * N, device=dev, generator=g) * 3).to(dt)
        sigma[::5] = 0
        sigma[7] = float("nan")
        mine = grid.clone()
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        total = R.sweep_update(mine, cells, sigma, 2.0, 0.95, counter)
        tmp = torch.full_like(grid, -1)
        tmp.index_reduce_(0, cells.long(), torch.nan_to_num(sigma.float() * 2.0, nan=-1.0), "amax", include_self=True)
        valid = (grid >= 0) & (tmp >= 0)
        nan_cell = int(cells[7])
        valid[nan_cell] = False  # the NaN wins its cell and leaves it unchanged, like `tmp >= 0` in the reference
        ref = torch.where(valid, torch.maximum(grid * 0.95, tmp), grid)
        assert torch.equal(mine, ref)
        assert bool((mine[:777] == -1).all()) and int(counter) == 1
        torch.testing.assert_close(total, ref.clamp(min=0).sum(), rtol=1e-5, atol=0)


def test_march_rays_train_makes_near_far_and_jitter_itself(hip, march_path):
    """build extension `aabb` of s3d_march_rays_train: near_far_from_aabb (+ the counter-based jitter) inside the marcher ==
    the two calls in sequence, bit for bit, on both kernel paths"""
    R = hip.RaymarchingBackend
    _, bits = _scene(seed=0)
    ro, rd = _rays(4096, seed=9)
    ro, rd, bits = ro.cuda(), rd.cuda(), bits.cuda()
    N, M = 4096, 4096 * 200
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device="cuda")
    step = torch.full((1,), 11, dtype=torch.int32, device="cuda")

    def buffers():
        return (torch.zeros(M, 3, device="cuda"), torch.zeros(M, 3, device="cuda"), torch.zeros(M, 2, device="cuda"),
                torch.empty(N, 3, dtype=torch.int32, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda"))
    nears, fars, noises = (torch.empty(N, device="cuda") for _ in range(3))
    R.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars, noises=noises, noise_step=step, noise_key=7)
    a = buffers()
    R.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, *a, noises)
    n2, f2, z2 = (torch.full((N,), float("nan"), device="cuda") for _ in range(3))
    b = buffers()
    R.march_rays_train(ro, rd, bits, 1.0, 0.0, 1024, N, 1, 128, M, n2, f2, *b, z2, aabb=aabb, min_near=0.2, noise_step=step,
                       noise_key=7)
    assert torch.equal(nears, n2) and torch.equal(fars, f2) and torch.equal(noises, z2)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert int(a[4][0]) > N


def test_sph_from_ray(oracle, hip):
    ro, rd = _rays(2048, seed=2)
    N = ro.shape[0]
    co = torch.empty(N, 2)
    cpu, gpu = _both(oracle, hip, "sph_from_ray", [ro * 0.1, rd, 4.0, N, co], 1)
    torch.testing.assert_close(gpu[4].cpu(), cpu[4], rtol=1e-5, atol=1e-5)  # atan2f: FP tolerance


def _march_train_args(ro, rd, bits, C, bound, M, perturb, dt_gamma=0.0, max_steps=1024, min_near=0.2, seed=0):
    from oracle import oracle_backend as ob
    N = ro.shape[0]
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
    nears, fars = torch.empty(N), torch.empty(N)
    ob.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, N, min_near, nears, fars)
    g = torch.Generator().manual_seed(seed)
    noises = torch.rand(N, generator=g) if perturb else torch.zeros(N)
    xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
    rays = torch.empty(N, 3, dtype=torch.int32)
    counter = torch.zeros(2, dtype=torch.int32)
    return [ro, rd, bits, bound, dt_gamma, max_steps, N, C, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises]


@pytest.fixture(params=[1, 2, 3], ids=["lane-per-ray", "wave-per-ray", "wave-per-ray-general"])
def march_path(request, hip):
    hip.RaymarchingBackend.set_march_path(request.param)
    yield request.param
    hip.RaymarchingBackend.set_march_path(0)


@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("cascade,bound,dt_gamma,max_steps", [(1, 1.0, 0.0, 1024), (2, 2.0, 1.0 / 128, 1024), (3, 4.0, 0.0, 4096),
                                                              (1, 1.0, 0.0, 20)])
def test_march_rays_train_bit_exact(oracle, hip, march_path, perturb, cascade, bound, dt_gamma, max_steps):
    _, bits = _scene(seed=0, cascade=cascade, bound=bound)
    ro, rd = _rays(4096, seed=3)
    args = _march_train_args(ro, rd, bits, cascade, bound, 4096 * 256, perturb, dt_gamma=dt_gamma, max_steps=max_steps)
    cpu, gpu = _both(oracle, hip, "march_rays_train", args, 5)
    # compaction: per-ray (id, offset, count) and the two counters, bit-exact
    assert torch.equal(cpu[15], gpu[15].cpu())
    assert torch.equal(cpu[16], gpu[16].cpu())
    m = int(cpu[16][0])
    assert m > (4096 * 3 if max_steps > 20 else 4096), "scene should produce samples"
    if max_steps == 20:
        assert int(cpu[15][:, 2].max()) == 20, "the per-ray max_steps cap must bind"
    for k in (12, 13, 14):  # xyzs, dirs, deltas bit-exact, including the untouched zero tail
        assert np.array_equal(cpu[k].numpy().view(np.uint32), gpu[k].cpu().numpy().view(np.uint32))


def test_march_rays_train_capped_budget_and_empty(oracle, hip, march_path):
    _, bits = _scene(seed=0)
    ro, rd = _rays(4096, seed=4)
    # budget smaller than the demand: rays whose span does not fit are dropped, nothing is written past M
    args = _march_train_args(ro, rd, bits, 1, 1.0, 20000, True)
    cpu, gpu = _both(oracle, hip, "march_rays_train", args, 5)
    assert torch.equal(cpu[15], gpu[15].cpu()) and torch.equal(cpu[16], gpu[16].cpu())
    assert int(cpu[16][0]) > 20000
    for k in (12, 13, 14):
        assert np.array_equal(cpu[k].numpy().view(np.uint32), gpu[k].cpu().numpy().view(np.uint32))
    # empty bitfield: every ray has zero samples
    args = _march_train_args(ro[:100].contiguous(), rd[:100].contiguous(), torch.zeros_like(bits), 1, 1.0, 1024, False)
    cpu, gpu = _both(oracle, hip, "march_rays_train", args, 5)
    assert torch.equal(cpu[15], gpu[15].cpu()) and int(gpu[16][0]) == 0 and int(gpu[16][1]) == 100
    # ragged N (not a multiple of 64) and N = 1
    for n in (1, 63, 65, 1000):
        args = _march_train_args(ro[:n].contiguous(), rd[:n].contiguous(), bits, 1, 1.0, n * 256, True)
        cpu, gpu = _both(oracle, hip, "march_rays_train", args, 5)
        assert torch.equal(cpu[15], gpu[15].cpu()) and torch.equal(cpu[16], gpu[16].cpu())


@pytest.mark.parametrize("M", [4096 * 256, 20000], ids=["roomy", "capped-budget"])
def test_training_kernels_zero_the_unfilled_rows_they_expose(hip, march_path, M):
    """seal3d_hip.h: with garbage (NaN) in the caller's buffers, march_rays_train and composite_rays_train_backward leave
    every row a consumer bounded by the device-side count can read — [0, min(round_up(total, 128), M)) — exactly as with
    zero-filled buffers (samples, zero pads, the dropped straddling ray's rows, samples behind an early termination)."""
    R = hip.RaymarchingBackend
    _, bits = _scene(seed=0)
    ro, rd = _rays(4096, seed=4)
    base = [a.cuda() if torch.is_tensor(a) else a for a in _march_train_args(ro, rd, bits, 1, 1.0, M, True)]
    dirty = list(base)
    for k in (12, 13, 14):
        dirty[k] = torch.full_like(base[k], float("nan"))
    dirty[15], dirty[16] = torch.empty_like(base[15]), torch.zeros_like(base[16])
    R.march_rays_train(*base)
    R.march_rays_train(*dirty)
    total = int(base[16][0])
    assert (total > M) == (M == 20000)
    end = min((total + 127) // 128 * 128, M)
    assert torch.equal(base[15], dirty[15]) and torch.equal(base[16], dirty[16])
    for k in (12, 13, 14):
        assert torch.equal(base[k][:end], dirty[k][:end])
        assert end == M or torch.isnan(dirty[k][end:]).all()  # nothing written behind the exposed rows
    # gradients: dense sigmas (early terminations), NaN-filled gradient buffers
    N, rays, deltas = 4096, base[15], base[14]
    g = torch.Generator().manual_seed(1)
    sigmas = torch.exp(torch.randn(M, generator=g) * 2 + 1).cuda()
    rgbs = torch.rand(M, 3, generator=g).cuda()
    ws, dp, im = (torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda"))
    for cpath in (0, 1):
        R.set_composite_path(cpath)
        R.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, 1e-4, ws, dp, im)
        gws, gim = torch.randn(N, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda()
        out = []
        for fill in (0.0, float("nan")):
            gs, gc = torch.full((M,), fill, device="cuda"), torch.full((M, 3), fill, device="cuda")
            R.composite_rays_train_backward(gws, gim, sigmas, rgbs, deltas, rays, ws, im, M, N, 1e-4, gs, gc)
            out.append((gs, gc))
        assert torch.equal(out[0][0][:end], out[1][0][:end]) and torch.equal(out[0][1][:end], out[1][1][:end])
        assert (out[0][0][:end] == 0).any() and (out[0][0][:end] != 0).any()
    R.set_composite_path(0)


def _composite_inputs(oracle, seed=0, n_rays=4096):
    _, bits = _scene(seed=0)
    ro, rd = _rays(n_rays, seed=seed)
    args = _march_train_args(ro, rd, bits, 1, 1.0, n_rays * 256, True)
    oracle.RaymarchingBackend.march_rays_train(*args)
    m = int(args[16][0])
    g = torch.Generator().manual_seed(seed + 10)
    # mix of dense (early-terminating) and thin samples
    sigmas = torch.exp(torch.randn(m, generator=g) * 2 + 1).contiguous()
    rgbs = torch.rand(m, 3, generator=g)
    return sigmas, rgbs, args[14][:m].contiguous(), args[15], m, n_rays


@pytest.mark.parametrize("cpath", [0, 1], ids=["wave-per-ray", "lane-per-ray"])
def test_composite_rays_train_forward_backward(oracle, hip, cpath):
    hip.RaymarchingBackend.set_composite_path(cpath)
    sigmas, rgbs, deltas, rays, M, N = _composite_inputs(oracle)
    ws, dp, im = torch.empty(N), torch.empty(N), torch.empty(N, 3)
    cpu, gpu = _both(oracle, hip, "composite_rays_train_forward", [sigmas, rgbs, deltas, rays, M, N, 1e-4, ws, dp, im], 3)
    for k in (7, 8, 9):
        torch.testing.assert_close(gpu[k].cpu(), cpu[k], rtol=1e-4, atol=1e-6)
    assert (cpu[7] == 0).any(), "expect some empty rays"
    g = torch.Generator().manual_seed(7)
    gws, gim = torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    gs, gc = torch.zeros(M), torch.zeros(M, 3)
    cpu2, gpu2 = _both(oracle, hip, "composite_rays_train_backward",
                       [gws, gim, sigmas, rgbs, deltas, rays, cpu[7], cpu[9], M, N, 1e-4, gs, gc], 2)
    torch.testing.assert_close(gpu2[12].cpu(), cpu2[12], rtol=1e-4, atol=1e-6)
    # grad_sigmas has cancellation; compare against the per-ray scale
    err = (gpu2[11].cpu() - cpu2[11]).abs().max() / cpu2[11].abs().max()
    assert err < 1e-4
    # early termination: samples after the cut keep the caller's zeros on both sides.  The cut is decided by
    # `T < T_thresh` on an exp() that differs in the last bits (__expf vs expf), so a ray may stop one sample
    # earlier or later: allow a 1e-3 fraction of samples to differ in written/unwritten state.
    mism = ((gpu2[11].cpu() == 0) != (cpu2[11] == 0)).float().mean()
    hip.RaymarchingBackend.set_composite_path(0)
    assert mism < 1e-3


def test_march_and_composite_inference_trace(oracle, hip):
    """three iterations of the run_cuda eval loop (nerf/renderer.py:341-367), incl. the kill pattern"""
    _, bits = _scene(seed=0)
    N = 8192
    ro, rd = _rays(N, seed=6)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    nears, fars = torch.empty(N), torch.empty(N)
    oracle.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
    st = {}
    for dev in ("cpu", "cuda"):
        be = oracle.RaymarchingBackend if dev == "cpu" else hip.RaymarchingBackend
        t = lambda x: x.to(dev)
        alive = torch.arange(N, dtype=torch.int32, device=dev)
        rays_t = t(nears).clone()
        ws, dp, im = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
        trace = []
        for it in range(3):
            n_alive = alive.shape[0]
            n_step = max(min(N // n_alive, 8), 1)
            M = n_alive * n_step
            M += 128 - M % 128
            xyzs, dirs, deltas = (torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev))
            noises = torch.zeros(n_alive, device=dev)
            be.march_rays(n_alive, n_step, alive, rays_t, t(ro), t(rd), 1.0, 0.0, 1024, 1, 128, t(bits), t(nears), t(fars),
                          xyzs, dirs, deltas, noises)
            # deterministic pseudo-network: density from the analytic box scene, colour from position
            lo, hi = syn.lego_like_boxes(0)
            sig = syn.box_density(xyzs.cpu(), lo, hi).to(dev) * 4
            rgb = (xyzs * 0.5 + 0.5).clamp(0, 1)
            be.composite_rays(n_alive, n_step, 1e-2, alive, rays_t, sig, rgb, deltas, ws, dp, im)
            trace.append((xyzs.cpu(), deltas.cpu(), alive.cpu().clone()))
            if dev == "cuda":
                # device-side compaction (wave ballot) must equal the host boolean-mask compaction
                comp, cnt = torch.empty_like(alive), torch.empty(1, dtype=torch.int32, device=dev)
                be.compact_alive(alive, n_alive, comp, cnt)
                ref = alive[alive >= 0]
                assert int(cnt) == ref.shape[0] and torch.equal(comp[:int(cnt)], ref)
            alive = alive[alive >= 0]
        st[dev] = (trace, ws.cpu(), dp.cpu(), im.cpu(), rays_t.cpu())
    for (xa, da, aa), (xb, db, ab) in zip(st["cpu"][0], st["cuda"][0]):
        assert np.array_equal(xa.numpy().view(np.uint32), xb.numpy().view(np.uint32))
        assert np.array_equal(da.numpy().view(np.uint32), db.numpy().view(np.uint32))
        assert torch.equal(aa, ab), "kill pattern differs"
    for k in (1, 2, 3, 4):
        torch.testing.assert_close(st["cuda"][k], st["cpu"][k], rtol=1e-4, atol=1e-6)
    assert (st["cpu"][0][2][2] == -1).any()


@pytest.mark.parametrize("cascade,bound,dt_gamma,n_alive,n_step", [
    (1, 1.0, 0.0, 640000, 4),        # first iterations of an 800x800 frame: every ray alive (wave pools of 128 slots)
    (1, 1.0, 0.0, 300000, 9),        # later iteration: a subset alive, rays_t advanced into the scene
    (2, 2.0, 1 / 128, 290000, 7),    # two cascades, growing steps (the general mip-level path)
    (1, 1.0, 0.0, 1000, 32),         # tail of the loop
])
def test_march_rays_full_frame_bit_exact(oracle, hip, cascade, bound, dt_gamma, n_alive, n_step):
    """s3d_march_rays at frame size (two launches: t walk with lane refill from wave pools, then row expansion) vs the
    oracle's per-ray loop: every sample row bit for bit, unfilled slots and padding rows zero although the buffers arrive
    as garbage (zero_unfilled), device-side alive count included."""
    _, bits = _scene(seed=0, cascade=cascade, bound=bound)
    N = 800 * 800
    poses = syn.orbit_poses(1, seed=9)
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800)
    ro, rd = r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous()
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound])
    nears, fars = torch.empty(N), torch.empty(N)
    oracle.RaymarchingBackend.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
    g = torch.Generator().manual_seed(11)
    alive = torch.randperm(N, generator=g)[:n_alive].sort().values.int()
    rays_t = nears + torch.rand(N, generator=g) * (0.0 if n_alive == N else 1.5)  # (later iterations start inside the scene)
    noises = torch.rand(n_alive, generator=g)
    M = n_alive * n_step
    M += 128 - M % 128
    H = 128
    x0, d0, l0 = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
    oracle.RaymarchingBackend.march_rays(n_alive, n_step, alive, rays_t, ro, rd, bound, dt_gamma, 1024, cascade, H, bits, nears,
                                         fars, x0, d0, l0, noises)
    c = lambda t: t.cuda()
    for dev_count in (False, True):
        # device-side count: the launch is sized for an upper bound, the real count sits in device memory
        bound_n = n_alive if not dev_count else min(N, n_alive + 4097)
        alive_g = torch.full((bound_n,), -7, dtype=torch.int32)
        alive_g[:n_alive] = alive
        Mg = max(M, bound_n * n_step + 128)
        xg, dg, lg = (torch.full((Mg, k), float("nan"), device="cuda") for k in (3, 3, 2))
        cnt = torch.tensor([n_alive], dtype=torch.int32, device="cuda") if dev_count else None
        rows = torch.zeros(1, dtype=torch.int32, device="cuda") if dev_count else None
        noise_g = torch.zeros(bound_n)
        noise_g[:n_alive] = noises
        hip.RaymarchingBackend.march_rays(bound_n, n_step, c(alive_g), c(rays_t), c(ro), c(rd), bound, dt_gamma, 1024, cascade, H,
                                          c(bits), c(nears), c(fars), xg, dg, lg, c(noise_g), n_alive_dev=cnt, n_rows_out=rows,
                                          zero_unfilled=True)
        torch.cuda.synchronize()
        if dev_count:
            assert int(rows) == n_alive * n_step
        live = n_alive * n_step
        end = ((live + 127) // 128) * 128 if dev_count else Mg   # rows a count-bounded consumer may read / the whole buffer
        for a, b, k in ((x0, xg, 3), (d0, dg, 3), (l0, lg, 2)):
            assert np.array_equal(a[:live].numpy().view(np.uint32), b[:live].cpu().numpy().view(np.uint32))
            assert float(b[live:end].abs().sum()) == 0.0, "padding rows must be zero"
    assert (l0[:n_alive * n_step, 0] == 0).any() and (l0[:, 0] != 0).any()


@pytest.mark.parametrize("dtype,n_step", [(torch.float32, 12), (torch.float32, 11), (torch.float16, 12)])
def test_composite_rays_four_samples_per_trip_is_the_scalar_loop(oracle, hip, dtype, n_step):
    """composite_rays reads a ray's rows four samples per trip (fp32: multi-dword loads on the rows' natural alignment, any
    n_step, the n_step % 4 tail one by one; fp16: 8-byte loads when n_step is a multiple of four and the rows are aligned —
    a buffer offset by one element takes the one-sample loop): same bits either way, the oracle's kill pattern and values."""
    g = torch.Generator().manual_seed(3)
    n_alive, N = 5000, 6000
    alive = torch.randperm(N, generator=g)[:n_alive].int()
    sig = torch.rand(n_alive * n_step, generator=g) * 30
    rgb = torch.rand(n_alive * n_step, 3, generator=g)
    deltas = torch.rand(n_alive * n_step, 2, generator=g) * 0.01 + 1e-3
    cut = torch.randint(0, n_step + 1, (n_alive,), generator=g)           # slots a ray did not fill: deltas == 0
    fill = (torch.arange(n_step)[None, :] < cut[:, None]).reshape(-1)
    deltas[~fill] = 0
    state = lambda dev: (alive.clone().to(dev), torch.rand(N, generator=torch.Generator().manual_seed(5)).to(dev) + 0.2,
                         torch.rand(N, generator=torch.Generator().manual_seed(6)).to(dev) * 0.5, torch.zeros(N, device=dev),
                         torch.zeros(N, 3, device=dev))
    a0, t0, w0, d0, i0 = state("cpu")
    oracle.RaymarchingBackend.composite_rays(n_alive, n_step, 1e-2, a0, t0, sig, rgb, deltas, w0, d0, i0)
    outs = []
    for misalign in (0, 1):
        a, t, w, d, im = state("cuda")
        sg = torch.zeros(sig.numel() + 4, dtype=dtype, device="cuda")[misalign: misalign + sig.numel()]
        sg.copy_(sig)
        cg = torch.zeros(rgb.numel() + 4, dtype=dtype, device="cuda")[misalign: misalign + rgb.numel()].view(-1, 3)
        cg.copy_(rgb)
        hip.RaymarchingBackend.composite_rays(n_alive, n_step, 1e-2, a, t, sg, cg, deltas.cuda(), w, d, im)
        outs.append((a.cpu(), t.cpu(), w.cpu(), d.cpu(), im.cpu()))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y), "vector and scalar loops differ"
    if dtype == torch.float32:
        assert torch.equal(outs[0][0], a0), "kill pattern differs from the oracle"
        for x, y in zip(outs[0][1:], (t0, w0, d0, i0)):
            torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6)


def test_full_size_properties(hip):
    """BASELINE size (800x800 rays): size-independent properties instead of an oracle run."""
    _, bits = _scene(seed=0)
    N = 800 * 800
    poses = syn.orbit_poses(1, seed=9)
    r = syn.get_rays(poses, syn.lego_intrinsics(), 800, 800)
    ro, rd = r["rays_o"][0].contiguous().cuda(), r["rays_d"][0].contiguous().cuda()
    be = hip.RaymarchingBackend
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1]).cuda()
    nears, fars = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    be.near_far_from_aabb(ro, rd, aabb, N, 0.2, nears, fars)
    M = 40_000_000
    xyzs, dirs, deltas = torch.zeros(M, 3, device="cuda"), torch.zeros(M, 3, device="cuda"), torch.zeros(M, 2, device="cuda")
    rays = torch.empty(N, 3, dtype=torch.int32, device="cuda")
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    be.march_rays_train(ro, rd, bits.cuda(), 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                        torch.zeros(N, device="cuda"))
    rays_c = rays.cpu().long()
    total = int(counter[0])
    assert int(counter[1]) == N and total <= M
    # spans are the exclusive prefix sum of the counts, in ray order
    assert torch.equal(rays_c[:, 0], torch.arange(N))
    assert torch.equal(rays_c[:, 1], torch.cumsum(rays_c[:, 2], 0) - rays_c[:, 2])
    assert int(rays_c[:, 2].sum()) == total
    # every written sample lies in an occupied cell; nothing written past `total`
    pts = xyzs[:total]
    cell = ((pts * 0.5 + 0.5) * 128).clamp(0, 127).int()
    idx = torch.empty(total, dtype=torch.int32, device="cuda")
    be.morton3D(cell.contiguous(), total, idx)
    b = bits.cuda()[(idx.long() >> 3)]
    assert bool(((b >> (idx & 7).to(torch.uint8)) & 1).all())
    assert float(xyzs[total:].abs().sum()) == 0.0
    # deltas: dt == dt_min everywhere (dt_gamma = 0)
    assert bool((deltas[:total, 0] == deltas[0, 0]).all())


@pytest.mark.parametrize("with_depth", [False, True], ids=["mse", "mse+depth"])
@pytest.mark.parametrize("budget", ["fits", "short"], ids=["all-rays-fit", "budget-cuts-rays"])
def test_composite_loss_one_launch_is_the_three_launch_sequence_bit_for_bit(oracle, hip, with_depth, budget):
    """s3d_composite_rays_train_loss = composite forward -> bg_mse_forward(announced gradient) -> composite backward: pixel, loss,
    loss gradients and sample gradients, incl. empty rays, early terminations, rays behind a too small budget M and NaN-filled
    gradient buffers (the rows a count-bounded consumer reads are written by the kernel itself)"""
    R, H = hip.RaymarchingBackend, hip.NgpHeadBackend
    sigmas, rgbs, deltas, rays, m, N = _composite_inputs(oracle, seed=3)
    M = m if budget == "fits" else (m * 3 // 4) // 128 * 128  # rays whose span ends behind M are dropped (raymarching.cu:416)
    end = M
    sigmas, rgbs, deltas = sigmas[:M].contiguous().cuda(), rgbs[:M].contiguous().cuda(), deltas[:M].contiguous().cuda()
    rays = rays.cuda()
    g = torch.Generator().manual_seed(11)
    gt = torch.rand(N, 3, generator=g).cuda()
    gt_depth = (torch.rand(N, generator=g) * 3).cuda() if with_depth else None
    scale = torch.tensor(1024.0, device="cuda")
    bg = (1.0, 0.5, 0.25)
    # the three launches
    ws, dp, im = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
    R.set_composite_path(0)
    R.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, 1e-4, ws, dp, im)
    loss, gi, gw = torch.empty((), device="cuda"), torch.empty(N, 3, device="cuda"), torch.empty(N, device="cuda")
    H.bg_mse_forward(im, ws, gt, bg, loss, scale, gi, gw, **(dict(depth=dp, gt_depth=gt_depth, depth_weight=0.7) if with_depth else {}))
    rows = sigmas.shape[0]
    gs, gc = torch.full((rows,), float("nan"), device="cuda"), torch.full((rows, 3), float("nan"), device="cuda")
    R.composite_rays_train_backward(gw, gi, sigmas, rgbs, deltas, rays, ws, im, M, N, 1e-4, gs, gc)
    # the one launch (twice through the same scratch)
    work = torch.full((4 * N,), float("nan"), device="cuda")
    for _ in range(2):
        ws2, dp2, im2 = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
        loss2, gi2, gw2 = torch.empty((), device="cuda"), torch.empty(N, 3, device="cuda"), torch.empty(N, device="cuda")
        gs2, gc2 = torch.full((rows,), float("nan"), device="cuda"), torch.full((rows, 3), float("nan"), device="cuda")
        R.composite_rays_train_loss(sigmas, rgbs, deltas, rays, M, N, 1e-4, gt, bg, scale, ws2, dp2, im2, gs2, gc2, loss2, work,
                                    gt_depth=gt_depth, depth_weight=0.7, grad_image=gi2, grad_weights_sum=gw2)
        for a, b, name in ((ws, ws2, "weights_sum"), (dp, dp2, "depth"), (im, im2, "image"), (loss, loss2, "loss"), (gi, gi2, "grad_image"),
                           (gw, gw2, "grad_weights_sum"), (gs, gs2, "grad_sigmas"), (gc, gc2, "grad_rgbs")):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name
    assert torch.isfinite(gs[:end]).all() and (gs[:end] != 0).any() and (ws == 0).any()
    if budget == "short":
        assert int(((rays[:, 1] + rays[:, 2]) > M).sum()) > 0, "expect rays behind the budget"


def test_composite_loss_function_backward_with_any_other_gradient_is_the_unfused_autograd(hip, oracle):
    """raymarching.composite_rays_train_loss hands out its precomputed gradients only for the announced upstream gradient; another
    gradient of the loss, or a gradient of the pixel, goes through the unfused kernels — and equals composite_rays_train + _BgMse"""
    import raymarching
    from nerf.trainer import _BgMse
    sigmas, rgbs, deltas, rays, M, N = _composite_inputs(oracle, seed=5, n_rays=1024)
    sigmas, rgbs, deltas, rays = sigmas.cuda(), rgbs.cuda(), deltas.cuda(), rays.cuda()
    gt = torch.rand(N, 3, generator=torch.Generator().manual_seed(2)).cuda()
    bg = (1.0, 1.0, 1.0)
    scale = torch.tensor(512.0, device="cuda")
    from raymarching import raymarching as rm
    prev = rm._backend
    rm._backend = hip.RaymarchingBackend
    try:
        def grads(fused, root, extra):
            s, c = sigmas.clone().requires_grad_(True), rgbs.clone().requires_grad_(True)
            if fused:
                loss, ws, dp, im = raymarching.composite_rays_train_loss(s, c, deltas, rays, 1e-4, gt, bg, scale)
            else:
                ws, dp, im = raymarching.composite_rays_train(s, c, deltas, rays, 1e-4)
                loss = _BgMse.apply(im, ws, gt, bg, scale)
            total = loss * 1.0 if not extra else loss + (im * 0.25).sum() + ws.sum() * 0.5
            total.backward(gradient=root)
            return loss.detach(), s.grad, c.grad
        for root, extra in ((scale, False), (torch.tensor(3.0, device="cuda"), False), (scale, True)):
            a, b = grads(True, root, extra), grads(False, root, extra)
            assert torch.equal(a[0], b[0])
            if root is scale and not extra:  # (loss * 1.0 hands the root gradient on as a new tensor: the general path on both sides)
                pass
            torch.testing.assert_close(a[1], b[1], rtol=1e-5, atol=1e-6 * float(b[1].abs().max()))
            torch.testing.assert_close(a[2], b[2], rtol=1e-5, atol=1e-6 * float(b[2].abs().max()))
        # the announced gradient itself: bit for bit
        s, c = sigmas.clone().requires_grad_(True), rgbs.clone().requires_grad_(True)
        loss, *_ = raymarching.composite_rays_train_loss(s, c, deltas, rays, 1e-4, gt, bg, scale)
        loss.backward(gradient=scale)
        s2, c2 = sigmas.clone().requires_grad_(True), rgbs.clone().requires_grad_(True)
        ws, dp, im = raymarching.composite_rays_train(s2, c2, deltas, rays, 1e-4)
        _BgMse.apply(im, ws, gt, bg, scale).backward(gradient=scale)
        assert torch.equal(s.grad, s2.grad) and torch.equal(c.grad, c2.grad)
    finally:
        rm._backend = prev
