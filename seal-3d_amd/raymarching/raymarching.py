"""raymarching — drop-in for the reference's `raymarching` package on MI355X.

Same module-level callables, argument meaning, defaults, return values and
AMP behaviour as raymarching/raymarching.py of the reference (cited per op);
the native work goes to libseal3d_hip through `_backend`
(s3d_hip.RaymarchingBackend).  Tests may swap `_backend` for the CPU oracle;
the product path never does.
"""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import s3d_hip

_backend = s3d_hip.RaymarchingBackend


def _empty_332(M, dtype, device):
    flat = torch.empty(M * 8, dtype=dtype, device=device)
    return flat[:3 * M].view(M, 3), flat[3 * M:6 * M].view(M, 3), flat[6 * M:].view(M, 2)


def _zeros_332(M, dtype, device):
    """xyzs [M,3], dirs [M,3], deltas [M,2], zero-initialised as in the reference (raymarching.py:205-207, padding rows
    must read as zeros) — three contiguous tensors carved out of ONE filled buffer (one fill kernel instead of three)."""
    flat = torch.zeros(M * 8, dtype=dtype, device=device)
    return flat[:3 * M].view(M, 3), flat[3 * M:6 * M].view(M, 3), flat[6 * M:].view(M, 2)

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train", "composite_rays_train_loss", "march_rays", "composite_rays", "compact_rays_alive"]


def _on_device(t):
    """The reference wrappers move host tensors to the GPU (`if not x.is_cuda: x = x.cuda()`,
    raymarching.py:34-35).  Same here whenever the active backend is the GPU one."""
    if getattr(_backend, "device_type", "cuda") == "cuda" and not t.is_cuda:
        return t.cuda()
    return t


def _rays(rays_o, rays_d):
    rays_o = _on_device(rays_o).contiguous().view(-1, 3)
    rays_d = _on_device(rays_d).contiguous().view(-1, 3)
    return rays_o, rays_d


def _align_up(m, align):
    # raymarching.py:200 — `m += align - m % align` (adds a full `align` when already aligned)
    return m + align - m % align if align > 0 else m


class _NearFarFromAABB(Function):
    """raymarching.py:19-49"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2, noise_step=None, noise_key=0):
        """`noise_step` (build extension, seal3d_hip.h): int32 GPU tensor holding the running step number — a third output,
        the per-ray jitter for march_rays_train(noises=...), is drawn by the same kernel"""
        rays_o, rays_d = _rays(rays_o, rays_d)
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        if noise_step is None:
            _backend.near_far_from_aabb(rays_o, rays_d, aabb.contiguous(), N, min_near, nears, fars)
            return nears, fars
        noises = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        _backend.near_far_from_aabb(rays_o, rays_d, aabb.contiguous(), N, min_near, nears, fars, noises=noises,
                                    noise_step=noise_step, noise_key=noise_key)
        return nears, fars, noises


near_far_from_aabb = _NearFarFromAABB.apply


class _SphFromRay(Function):
    """raymarching.py:52-80"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, radius):
        rays_o, rays_d = _rays(rays_o, rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        _backend.sph_from_ray(rays_o, rays_d, radius, N, coords)
        return coords


sph_from_ray = _SphFromRay.apply


class _Morton3D(Function):
    """raymarching.py:83-103"""

    @staticmethod
    def forward(ctx, coords):
        coords = _on_device(coords)
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        _backend.morton3D(coords.int().contiguous(), N, indices)
        return indices


morton3D = _Morton3D.apply


class _Morton3DInvert(Function):
    """raymarching.py:105-126"""

    @staticmethod
    def forward(ctx, indices):
        indices = _on_device(indices)
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        _backend.morton3D_invert(indices.int().contiguous(), N, coords)
        return coords


morton3D_invert = _Morton3DInvert.apply


class _Packbits(Function):
    """raymarching.py:129-155"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, grid, thresh, bitfield=None):
        grid = _on_device(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        _backend.packbits(grid, N, thresh, bitfield)
        return bitfield


packbits = _Packbits.apply


class _MarchRaysTrain(Function):
    """raymarching.py:161-235.  Sample spans are packed in ray order (deterministic)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, trim=True, noises=None,
                zero_fill=True, aabb=None, min_near=0.2, noise_step=None, noise_key=0):
        """`trim=False` (build extension): without a sample budget the reference reads the sample count back and trims the
        N * max_steps buffers (raymarching.py:223-231: a device->host sync).  A caller whose whole sample path takes the
        device-side count (`n_valid`, seal3d_hip.h) keeps the full buffers instead: no sync, static shapes.
        `noises` (build extension): the per-ray jitter [N] in [0, 1) when the caller has drawn it already (perturb).
        `zero_fill=False` (build extension, HIP backend): the reference zero-fills the M-row buffers; the HIP kernels write
        zeros to every unfilled row below the sample count rounded up to 128 themselves (seal3d_hip.h), which is all a caller
        reads whose sample path takes the device-side count — it can skip the fill.
        `aabb` (build extension, HIP backend): near_far_from_aabb(aabb, min_near[, noise_step, noise_key]) is made by the
        marcher itself; `nears` / `fars` may then be None."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        density_bitfield = _on_device(density_bitfield).contiguous()
        dev, dt = rays_o.device, rays_o.dtype
        N = rays_o.shape[0]

        # sample budget: all rays x max_steps until a running mean exists (raymarching.py:193-203)
        M = N * max_steps
        budgeted = (not force_all_rays) and mean_count > 0
        if budgeted:
            M = _align_up(mean_count, align)

        xyzs, dirs, deltas = _zeros_332(M, dt, dev) if zero_fill else _empty_332(M, dt, dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        extra = {}
        if aabb is not None:
            nears = torch.empty(N, dtype=dt, device=dev)
            fars = torch.empty(N, dtype=dt, device=dev)
            extra = dict(aabb=aabb.contiguous(), min_near=min_near)
            if perturb and noises is None and noise_step is not None:
                noises = torch.empty(N, dtype=dt, device=dev)
                extra.update(noise_step=noise_step, noise_key=noise_key)
        if not perturb:
            noises = torch.zeros(N, dtype=dt, device=dev)
        elif noises is None:
            noises = torch.rand(N, dtype=dt, device=dev)

        _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M,
                                  nears.contiguous(), fars.contiguous(), xyzs, dirs, deltas, rays, step_counter, noises,
                                  **extra)

        if not budgeted and trim:
            # first iterations only: one D2H read to trim the over-allocation (raymarching.py:223-231)
            m = _align_up(int(step_counter[0].item()), align)
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _MarchRaysTrain.apply


class _CompositeRaysTrain(Function):
    """raymarching.py:238-291 — differentiable w.r.t. sigmas and rgbs (grad_depth is not propagated)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4, zero_grads=True):
        """`zero_grads=False` (build extension, HIP backend): as `zero_fill` of march_rays_train, for the gradient buffers"""
        ctx.zero_grads = zero_grads
        # the kernels are fp32 (the reference reaches them through autocast's cast_inputs); outside autocast a half-precision
        # network output (ffmlp always computes in fp16) is converted here instead of being misread
        sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        _backend.composite_rays_train_forward(sigmas, rgbs, deltas.contiguous(), rays, M, N, T_thresh, weights_sum,
                                              depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = (M, N, T_thresh)
        ctx.set_materialize_grads(False)  # depth takes no gradient (below): no zero tensor made for it every step
        return weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        if grad_weights_sum is None and grad_image is None:
            return None, None, None, None, None, None
        if grad_weights_sum is None:
            grad_weights_sum = torch.zeros_like(weights_sum)
        if grad_image is None:
            grad_image = torch.zeros_like(image)
        # zero-initialised like the reference (raymarching.py:283-284) — one fill for both
        alloc = torch.zeros if ctx.zero_grads else torch.empty
        flat = alloc(sigmas.numel() + rgbs.numel(), dtype=sigmas.dtype, device=sigmas.device)
        grad_sigmas = flat[:sigmas.numel()].view_as(sigmas)
        grad_rgbs = flat[sigmas.numel():].view_as(rgbs)
        _backend.composite_rays_train_backward(grad_weights_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs,
                                               deltas.contiguous(), rays, weights_sum, image, M, N, T_thresh,
                                               grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None, None


composite_rays_train = _CompositeRaysTrain.apply


class _CompositeRaysTrainLoss(Function):
    """Build extension (HIP backend): composite_rays_train, the background + MSE criterion of the training step
    (nerf/renderer.py:316 `image + (1 - weights_sum) * bg_color`, nerf/utils.py:484-489) and composite_rays_train's backward for
    the loss's ANNOUNCED upstream gradient `expected_grad` (the loss scale under GradScaler) in one launch, plus a one-workgroup
    sum of the loss terms (seal3d_hip.h: s3d_composite_rays_train_loss).  Returns (loss, weights_sum, depth, image).  backward() hands out the
    gradients the forward launch wrote when it receives exactly that tensor as the loss's gradient and nothing else; any other
    use takes the unfused kernels."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh, gt, bg, expected_grad, workspace=None, gt_depth=None, depth_weight=1.0,
                zero_grads=True):
        sigmas, rgbs, deltas = sigmas.float().contiguous(), rgbs.float().contiguous(), deltas.contiguous()
        gt = gt.float().contiguous().view(-1, 3)
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        # zero_grads=False: as for composite_rays_train — the kernel zeroes the rows a count-bounded consumer still reads itself
        flat = (torch.zeros if zero_grads else torch.empty)(4 * M, dtype=torch.float32, device=dev)
        grad_sigmas, grad_rgbs = flat[:M], flat[M:].view(M, 3)
        if workspace is None:
            workspace = torch.empty(4 * N, dtype=torch.float32, device=dev)
        if gt_depth is not None:
            gt_depth = gt_depth.float().contiguous().view(-1)
        _backend.composite_rays_train_loss(sigmas, rgbs, deltas, rays, M, N, T_thresh, gt, bg, expected_grad, weights_sum, depth,
                                           image, grad_sigmas, grad_rgbs, loss, workspace, gt_depth=gt_depth,
                                           depth_weight=float(depth_weight))
        ctx.pre = (grad_sigmas, grad_rgbs, expected_grad.data_ptr(), expected_grad._version)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image, gt)
        ctx.dims = (M, N, T_thresh, bg)
        ctx.set_materialize_grads(False)
        return loss, weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_loss, g_ws, g_depth, g_image):
        none = (None,) * 10
        if g_loss is None and g_ws is None and g_image is None:
            return (None, None) + none
        if (g_ws is None and g_image is None and g_loss is not None and g_loss.dtype == torch.float32
                and g_loss.data_ptr() == ctx.pre[2] and g_loss._version == ctx.pre[3]):
            return (ctx.pre[0], ctx.pre[1]) + none  # (the announced upstream gradient: already computed)
        sigmas, rgbs, deltas, rays, weights_sum, image, gt = ctx.saved_tensors
        M, N, T_thresh, bg = ctx.dims
        gi, gw = torch.zeros_like(image), torch.zeros_like(weights_sum)
        if g_loss is not None:
            _head().bg_mse_backward(image, weights_sum, gt, bg, g_loss.float().contiguous(), gi, gw)
        if g_image is not None:
            gi = gi + g_image
        if g_ws is not None:
            gw = gw + g_ws
        flat = torch.zeros(4 * M, dtype=torch.float32, device=sigmas.device)
        grad_sigmas, grad_rgbs = flat[:M], flat[M:].view(M, 3)
        _backend.composite_rays_train_backward(gw.contiguous(), gi.contiguous(), sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                               T_thresh, grad_sigmas, grad_rgbs)
        return (grad_sigmas, grad_rgbs) + none


def _head():
    import s3d_hip
    return s3d_hip.NgpHeadBackend


def composite_rays_train_loss(sigmas, rgbs, deltas, rays, T_thresh, gt, bg, expected_grad, workspace=None, gt_depth=None,
                              depth_weight=1.0, zero_grads=True):
    if not hasattr(_backend, "composite_rays_train_loss"):
        raise RuntimeError("composite_rays_train_loss: the active raymarching backend has no fused loss launch")
    return _CompositeRaysTrainLoss.apply(sigmas, rgbs, deltas, rays, T_thresh, gt, bg, expected_grad, workspace, gt_depth, depth_weight,
                                         zero_grads)


class _MarchRays(Function):
    """raymarching.py:297-348"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024, n_alive_dev=None, n_rows_out=None):
        """`n_alive_dev` / `n_rows_out` (build extension, sync-free loop): `n_alive` is then an upper bound, the int32 GPU
        tensor `n_alive_dev` holds the real count, and `n_rows_out` receives count * n_step (seal3d_hip.h)"""
        rays_o, rays_d = _rays(rays_o, rays_d)
        dev, dt = rays_o.device, rays_o.dtype
        M = _align_up(n_alive * n_step, align)
        extra = {} if n_alive_dev is None else {"n_alive_dev": n_alive_dev, "n_rows_out": n_rows_out}
        if getattr(_backend, "zero_fills_march_rays", False):
            # HIP backend: the kernel zeroes the slots it does not fill and takes `no noise` as a null pointer — the two fill
            # launches per iteration of the inference loop (zero-initialised outputs, zero noises) are gone
            xyzs, dirs, deltas = _empty_332(M, dt, dev)
            noises = torch.rand(n_alive, dtype=dt, device=dev) if perturb else None
            extra["zero_unfilled"] = True
        else:
            xyzs, dirs, deltas = _zeros_332(M, dt, dev)
            noises = torch.rand(n_alive, dtype=dt, device=dev) if perturb else torch.zeros(n_alive, dtype=dt, device=dev)
        _backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                            density_bitfield, near, far, xyzs, dirs, deltas, noises, **extra)
        return xyzs, dirs, deltas


march_rays = _MarchRays.apply


class _CompositeRays(Function):
    """raymarching.py:351-373 — accumulates into weights_sum/depth/image IN PLACE, marks dead rays with -1."""

    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                T_thresh=1e-2, n_alive_dev=None):
        # the reference casts every input to fp32 (custom_fwd(cast_inputs=float32), raymarching.py:354).  The HIP kernel reads
        # fp16 sigmas / rgbs as they leave the network and converts on load — the same values, two cast launches per iteration
        # of the inference loop less; everything else (and every backend without that ability) is cast as in the reference
        extra = {} if n_alive_dev is None else {"n_alive_dev": n_alive_dev}
        half_ok = getattr(_backend, "zero_fills_march_rays", False)

        def as_input(t):
            return t.contiguous() if (half_ok and t.dtype == torch.float16) else t.float().contiguous()
        _backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, as_input(sigmas), as_input(rgbs), deltas.float().contiguous(), weights_sum, depth, image, **extra)
        return tuple()


composite_rays = _CompositeRays.apply


def compact_rays_alive(rays_alive, n_alive, n_in_dev=None):
    """Device-side equivalent of `rays_alive[rays_alive >= 0]` (nerf/renderer.py:363): stable wave-ballot
    compaction.  Returns (compacted buffer, device int32 count); the caller decides when to read the count.
    `n_in_dev`: int32 GPU tensor with the number of meaningful entries when `n_alive` is only an upper bound."""
    out = torch.empty_like(rays_alive)
    n_out = torch.empty(1, dtype=torch.int32, device=rays_alive.device)
    if n_in_dev is None:
        _backend.compact_alive(rays_alive, n_alive, out, n_out)
    else:
        _backend.compact_alive(rays_alive, n_alive, out, n_out, n_in_dev=n_in_dev)
    return out, n_out
