"""s3d_hip — ctypes binding of libseal3d_hip.so (the MI355X product path).

This is the only module that touches the native library.  It exposes five
backend objects whose methods have the names and argument order of the
reference's pybind ``_backend`` modules (raymarching/src/bindings.cpp:6-17,
gridencoder/src/bindings.cpp:6-8, shencoder/src/bindings.cpp:6-7,
freqencoder/src/bindings.cpp:6-7, ffmlp/src/bindings.cpp:6-10), taking torch
tensors that live on the GPU, launching on torch's *current* HIP stream, and
raising ``RuntimeError`` on any non-zero return code.

There is NO CPU fallback: if the library is missing or a tensor is not on the
GPU the call fails loudly.
"""
import ctypes as C
import os
import subprocess
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S3D_HIP_LIB") or os.path.join(_HERE, "libseal3d_hip.so")  # override: kernel-tuning experiments
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")

F32, F16 = 0, 1

# every symbol include/seal3d_hip.h declares
EXPORTS = [
    "s3d_last_error", "s3d_version",
    "s3d_near_far_from_aabb", "s3d_sph_from_ray", "s3d_morton3D", "s3d_morton3D_invert", "s3d_mip_levels", "s3d_packbits",
    "s3d_march_rays_train_workspace_size", "s3d_march_rays_train",
    "s3d_sweep_draw", "s3d_sweep_update_workspace_size", "s3d_sweep_update",
    "s3d_composite_rays_train_forward", "s3d_composite_rays_train_backward", "s3d_composite_rays_train_loss",
    "s3d_march_rays", "s3d_composite_rays", "s3d_compact_alive_workspace_size", "s3d_compact_alive",
    "s3d_grid_level_scales", "s3d_grid_encode_forward", "s3d_grid_encode_forward_pair", "s3d_grid_corner_indices", "s3d_grid_encode_backward",
    "s3d_grid_encode_backward_workspace_size", "s3d_grid_encode_backward_control_size",
    "s3d_grad_total_variation",
    "s3d_sh_encode_forward", "s3d_sh_encode_backward", "s3d_freq_encode_forward", "s3d_freq_encode_backward", "s3d_freq_encode_pack_forward", "s3d_freq_encode_pack_backward",
    "s3d_ffmlp_forward", "s3d_ffmlp_inference", "s3d_ffmlp_ngp_pair_inference", "s3d_ffmlp_wgrad_reduce_pair", "s3d_ffmlp_backward_workspace_size", "s3d_ffmlp_backward",
    "s3d_ffmlp_fused_backward_supported",
    "s3d_ffmlp_allocate_splitk", "s3d_ffmlp_free_splitk",
    "s3d_grads_nonfinite", "s3d_adam_step", "s3d_adam_step_multi", "s3d_adam_advance", "s3d_scaler_update", "s3d_step_ring_push", "s3d_step_epilogue",
    "s3d_ngp_mid_forward", "s3d_ngp_mid_backward", "s3d_ngp_mid2_forward", "s3d_ngp_mid2_backward", "s3d_ngp_rgb_forward", "s3d_ngp_rgb_backward",
    "s3d_bg_mse_forward", "s3d_bg_mse_backward", "s3d_bg_targets", "s3d_l1_pair_workspace_size", "s3d_l1_pair_loss",
    "s3d_seal_bbox_map", "s3d_seal_map_color", "s3d_grid_encode_backward_adam", "s3d_vm_features_forward",
    "s3d_aabb_normalize", "s3d_weighted_abs_sum_workspace_size", "s3d_weighted_abs_sum", "s3d_pack_linear_chain", "s3d_unpack_linear_chain",
    "s3d_vm_backward_max_bins", "s3d_vm_backward_keys", "s3d_vm_backward_bins_workspace_size", "s3d_vm_backward_bins",
    "s3d_vm_backward_stage_bytes", "s3d_vm_transpose_factors",
    "s3d_vm_features_backward", "s3d_vm_color_forward", "s3d_vm_color_backward",
]


def build(force=False, jobs=8):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-s", f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"seal3d HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (needs hipcc). There is no CPU fallback for the product path.")
        l = C.CDLL(LIB_PATH)
        l.s3d_last_error.restype = C.c_char_p
        l.s3d_version.restype = C.c_char_p
        for name in ("s3d_march_rays_train_workspace_size", "s3d_compact_alive_workspace_size",
                     "s3d_ffmlp_backward_workspace_size", "s3d_grid_encode_backward_workspace_size",
                     "s3d_grid_encode_backward_control_size", "s3d_l1_pair_workspace_size",
                     "s3d_sweep_update_workspace_size", "s3d_vm_backward_bins_workspace_size", "s3d_vm_backward_stage_bytes",
                     "s3d_weighted_abs_sum_workspace_size"):
            getattr(l, name).restype = C.c_size_t
        l.s3d_vm_backward_max_bins.restype = C.c_uint32
        l.s3d_grid_level_scales.restype = None
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = lib().s3d_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("seal3d HIP backend needs GPU tensors (no CPU fallback in the product path)")
    if not t.is_contiguous():
        raise RuntimeError("seal3d HIP backend needs contiguous tensors")
    return C.c_void_p(t.data_ptr())


def _u(x):
    return C.c_uint32(int(x))


def _nv(t):
    """`n_valid` of seal3d_hip.h: None, or an int32 GPU tensor whose first element is the sample count"""
    if t is None:
        return C.c_void_p(0)
    if t.dtype != torch.int32 or not t.is_cuda:
        raise RuntimeError("n_valid must be an int32 GPU tensor (the ray marcher's counter)")
    return C.c_void_p(t.data_ptr())


def _live(t, B):
    """`live` of s3d_grid_encode_forward: None, or an fp32 GPU tensor [B, ...] whose first column marks live rows"""
    if t is None:
        return C.c_void_p(0), C.c_uint32(0)
    if t.dtype != torch.float32 or not t.is_cuda or t.shape[0] != B:
        raise RuntimeError("live must be an fp32 GPU tensor with one row per input row")
    return C.c_void_p(t.data_ptr()), C.c_uint32(t.stride(0) if t.dim() > 1 else 1)


# Inference counterpart of row_limit: the deltas tensor of the current march_rays chunk (zero rows = unused slots)
_LIVE_ROWS = None


class live_rows:
    def __init__(self, deltas):
        self.value = deltas

    def __enter__(self):
        global _LIVE_ROWS
        self.saved, _LIVE_ROWS = _LIVE_ROWS, self.value
        return self

    def __exit__(self, *exc):
        global _LIVE_ROWS
        _LIVE_ROWS = self.saved
        return False


def active_live_rows(B):
    t = _LIVE_ROWS
    if t is not None and t.shape[0] == int(B) and t.dtype == torch.float32 and t.is_cuda:
        return t
    return None


# The padded sample batch of the current training render: (counter tensor, rows of the padded buffers).  The renderer
# announces it around the network call; a network whose whole sample path is native picks it up with
# `active_row_limit(B)` and hands it to every kernel explicitly (forward AND backward), so the work follows the samples.
_ROW_LIMIT = None


class row_limit:
    def __init__(self, counter, rows):
        self.value = (counter, int(rows))

    def __enter__(self):
        global _ROW_LIMIT
        self.saved, _ROW_LIMIT = _ROW_LIMIT, self.value
        return self

    def __exit__(self, *exc):
        global _ROW_LIMIT
        _ROW_LIMIT = self.saved
        return False


def active_row_limit(B):
    """the announced counter when it describes a batch of exactly B rows, else None"""
    if _ROW_LIMIT is not None and _ROW_LIMIT[1] == int(B):
        return _ROW_LIMIT[0]
    return None


def _shadow2(shadows):
    """(planes_t, lines_t) pointer arrays of VmBackend.transpose_factors()' result, or two NULLs"""
    if shadows is None:
        return None, None
    ptr3 = C.c_void_p * 3
    return ptr3(*[t.data_ptr() for t in shadows[0]]), ptr3(*[t.data_ptr() for t in shadows[1]])


def _shadow1(shadows):
    """planes_t pointer array of VmBackend.transpose_factors()' result, or NULL"""
    return (_shadow2(shadows)[0],)


def _f(x):
    return C.c_float(float(x))


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise RuntimeError(f"seal3d HIP backend: unsupported dtype {t.dtype} (float32/float16 only)")


def _need(t, dtype, name):
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")


class _Workspace:
    """Per-device, per-stream scratch owned by the binding (grown on demand, reused).  While the stream is being captured
    into a HIP graph the scratch is a plain temporary instead: a cached tensor would come out of THAT graph's private memory
    pool and dangle for every later caller once the graph is destroyed (torch captures every graph on the same stream)."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes, device):
        n = max(int(nbytes), 1 << 16)
        if torch.cuda.is_current_stream_capturing():
            return torch.empty(n, dtype=torch.uint8, device=device)
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            if b is None and len(self.buf) >= 16:  # (streams come and go: drop the oldest entries)
                for k in list(self.buf)[:8]:
                    del self.buf[k]
            b = torch.empty(n, dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


_ws = _Workspace()


class _ControlBlocks:
    """Zero-filled, self-cleaning control blocks of the binned grid backward (include/seal3d_hip.h: `control`), one per
    device, allocated OUTSIDE graph captures only: a block first requested while the stream is capturing would live in
    that graph's private pool (see _Workspace), so the call then runs without one (the library clears its control words
    with a launch of its own, as it does for any caller that passes none).  One block per device, not per stream: a
    captured step replays on whatever stream its owner picks, and the binding's callers issue their grid backwards in
    stream order (two of them running concurrently on different streams of one device would have to bring their own)."""

    def __init__(self):
        self.buf = {}
        self.retired = []  # smaller blocks that were replaced: never freed — a captured graph may still hold their address

    def get(self, nbytes, device):
        if nbytes <= 0:
            return None
        b = self.buf.get(device.index)
        if b is not None and b.numel() >= nbytes:
            return b
        if torch.cuda.is_current_stream_capturing():
            return None
        if b is not None:
            self.retired.append(b)  # (all-zero between calls like every control block; a few hundred KB)
        b = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        self.buf[device.index] = b
        return b


_ctl = _ControlBlocks()


def level_scales(L, S, H):
    out = (C.c_float * int(L))()
    lib().s3d_grid_level_scales(_u(L), _f(S), _u(H), out)
    return list(out)


class RaymarchingBackend:
    """raymarching/src/raymarching.h:7-18"""

    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars, noises=None, noise_step=None, noise_key=0):
        _need(rays_o, torch.float32, "rays_o")
        if noises is not None:
            _need(noises, torch.float32, "noises")
            if noise_step is not None:
                _need(noise_step, torch.int32, "noise_step")
        _check(lib().s3d_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), _u(N), _f(min_near), _p(nears),
                                            _p(fars), _p(noises), _p(noise_step), _u(int(noise_key) & 0xFFFFFFFF), _stream()),
               "near_far_from_aabb")

    @staticmethod
    def sweep_draw(u_uniform, u_occupied, occ_csum, H, bound, half_cell, noise_key=0, noise_step=None):
        """cells [2N] int32 + jittered positions [2N, 3] of one cascade's occupancy sweep (seal3d_hip.h)"""
        _need(u_uniform, torch.float64, "u_uniform"); _need(u_occupied, torch.float64, "u_occupied")
        _need(occ_csum, torch.int32, "occ_csum")
        N = u_uniform.numel()
        if u_occupied.numel() != N or occ_csum.numel() != H ** 3:
            raise RuntimeError("sweep_draw: u_uniform / u_occupied are [N], occ_csum is [H^3]")
        if noise_step is not None:
            _need(noise_step, torch.int32, "noise_step")
        cells = torch.empty(2 * N, dtype=torch.int32, device=u_uniform.device)
        xyzs = torch.empty(2 * N, 3, dtype=torch.float32, device=u_uniform.device)
        _check(lib().s3d_sweep_draw(_p(u_uniform), _p(u_occupied), _p(occ_csum), _u(N), _u(H), _f(bound), _f(half_cell),
                                    _u(int(noise_key) & 0xFFFFFFFF), _p(noise_step), _p(cells), _p(xyzs), _stream()), "sweep_draw")
        return cells, xyzs

    @staticmethod
    def sweep_update(density_grid, cells, sigma, density_scale, decay, step_counter=None):
        """EMA-max update of one cascade's density grid [H^3] (a contiguous fp32 view, in place) from the samples; returns the
        device scalar sum(max(grid, 0)) (seal3d_hip.h)"""
        _need(density_grid, torch.float32, "density_grid"); _need(cells, torch.int32, "cells")
        if sigma.dtype not in (torch.float16, torch.float32) or sigma.numel() != cells.numel():
            raise RuntimeError("sweep_update: sigma must be fp16 / fp32, one per cell sample")
        if step_counter is not None:
            _need(step_counter, torch.int32, "step_counter")
        n_cells = density_grid.numel()
        ws = _ws.get(lib().s3d_sweep_update_workspace_size(_u(n_cells)), density_grid.device)
        out = torch.empty((), dtype=torch.float32, device=density_grid.device)
        _check(lib().s3d_sweep_update(_p(density_grid), _u(n_cells), _p(cells), _p(sigma), C.c_int(_dt(sigma)), _u(cells.numel()),
                                      _f(density_scale), _f(decay), _p(ws), C.c_size_t(ws.numel()), _p(out), _p(step_counter),
                                      _stream()), "sweep_update")
        return out

    @staticmethod
    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        _need(rays_o, torch.float32, "rays_o")
        _check(lib().s3d_sph_from_ray(_p(rays_o), _p(rays_d), _f(radius), _u(N), _p(coords), _stream()),
               "sph_from_ray")

    @staticmethod
    def morton3D(coords, N, indices):
        _need(coords, torch.int32, "coords")
        _check(lib().s3d_morton3D(_p(coords), _u(N), _p(indices), _stream()), "morton3D")

    @staticmethod
    def morton3D_invert(indices, N, coords):
        _need(indices, torch.int32, "indices")
        _check(lib().s3d_morton3D_invert(_p(indices), _u(N), _p(coords), _stream()), "morton3D_invert")

    @staticmethod
    def mip_levels(xyz, dt, H, Cc):
        """Test hook: (mip_from_pos [N], mip_from_dt [N]) of the marching kernels' cascade selection (raymarching.cu:42-54)."""
        _need(xyz, torch.float32, "xyz")
        _need(dt, torch.float32, "dt")
        N = xyz.shape[0]
        assert dt.shape[0] == N
        mp = torch.empty(N, dtype=torch.int32, device=xyz.device)
        md = torch.empty(N, dtype=torch.int32, device=xyz.device)
        _check(lib().s3d_mip_levels(_p(xyz), _p(dt), _u(N), _u(H), _u(Cc), _p(mp), _p(md), _stream()), "mip_levels")
        return mp, md

    @staticmethod
    def packbits(grid, N, density_thresh, bitfield):
        _need(grid, torch.float32, "grid")
        _check(lib().s3d_packbits(_p(grid), _u(N), _f(density_thresh), _p(bitfield), _stream()), "packbits")

    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, H, M, nears, fars, xyzs, dirs,
                         deltas, rays, counter, noises, aabb=None, min_near=0.0, noise_step=None, noise_key=0):
        """`aabb` (build extension): near_far_from_aabb is part of the call — nears / fars (and noises, with `noise_step`) are
        outputs (seal3d_hip.h)"""
        _need(rays_o, torch.float32, "rays_o")
        if aabb is not None:
            _need(aabb, torch.float32, "aabb"); _need(nears, torch.float32, "nears"); _need(fars, torch.float32, "fars")
        if noise_step is not None:
            _need(noise_step, torch.int32, "noise_step")
        nbytes = lib().s3d_march_rays_train_workspace_size(_u(N), _u(max_steps))
        ws = _ws.get(nbytes, rays_o.device)
        _check(lib().s3d_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), _f(bound), _f(dt_gamma), _u(max_steps),
                                          _u(N), _u(Cc), _u(H), _u(M), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                                          _p(deltas), _p(rays), _p(counter), _p(noises), _p(ws),
                                          C.c_size_t(ws.numel()), C.c_int(RaymarchingBackend._march_path), _p(aabb),
                                          _f(min_near), _p(noise_step), _u(int(noise_key) & 0xFFFFFFFF), _stream()),
               "march_rays_train")

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        for t, n in ((sigmas, "sigmas"), (rgbs, "rgbs"), (deltas, "deltas")):
            _need(t, torch.float32, n)
        _check(lib().s3d_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), _u(M), _u(N),
                                                      _f(T_thresh), _p(weights_sum), _p(depth), _p(image),
                                                      C.c_int(RaymarchingBackend._composite_path), _stream()),
               "composite_rays_train_forward")

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image,
                                      M, N, T_thresh, grad_sigmas, grad_rgbs):
        for t, n in ((grad_image, "grad_image"), (grad_weights_sum, "grad_weights_sum"), (sigmas, "sigmas"), (rgbs, "rgbs"),
                     (deltas, "deltas")):
            _need(t, torch.float32, n)
        _check(lib().s3d_composite_rays_train_backward(_p(grad_weights_sum), _p(grad_image), _p(sigmas), _p(rgbs),
                                                       _p(deltas), _p(rays), _p(weights_sum), _p(image), _u(M),
                                                       _u(N), _f(T_thresh), _p(grad_sigmas), _p(grad_rgbs),
                                                       C.c_int(RaymarchingBackend._composite_path), _stream()),
               "composite_rays_train_backward")

    @staticmethod
    def composite_rays_train_loss(sigmas, rgbs, deltas, rays, M, N, T_thresh, gt, bg_rgb, grad_loss, weights_sum, depth, image,
                                  grad_sigmas, grad_rgbs, loss, workspace, gt_depth=None, depth_weight=1.0, grad_image=None,
                                  grad_weights_sum=None):
        """composite forward + background / MSE loss (announced upstream gradient `grad_loss`) + composite backward of one ray batch
        in one launch + a one-workgroup sum of the loss terms (seal3d_hip.h); workspace: 4N floats of scratch"""
        for t, n in ((sigmas, "sigmas"), (rgbs, "rgbs"), (deltas, "deltas"), (gt, "gt"), (grad_loss, "grad_loss"), (loss, "loss"),
                     (workspace, "workspace"), (grad_sigmas, "grad_sigmas"), (grad_rgbs, "grad_rgbs")):
            _need(t, torch.float32, n)
        if gt.numel() != 3 * N or workspace.numel() < 4 * N or grad_sigmas.numel() < M or grad_rgbs.numel() < 3 * M:
            raise RuntimeError("composite_rays_train_loss: gt [N,3], workspace >= 4N floats, grad_sigmas [M], grad_rgbs [M,3]")
        if gt_depth is not None:
            _need(gt_depth, torch.float32, "gt_depth")
            if gt_depth.numel() != N:
                raise RuntimeError("composite_rays_train_loss: gt_depth holds one value per ray")
        if (grad_image is None) != (grad_weights_sum is None):
            raise RuntimeError("composite_rays_train_loss: grad_image and grad_weights_sum come together")
        if RaymarchingBackend._composite_path != 0:
            raise RuntimeError("composite_rays_train_loss: the fused launch exists for the wave-per-ray path only")
        bg = (C.c_float * 3)(*[float(v) for v in bg_rgb])
        _check(lib().s3d_composite_rays_train_loss(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), _u(M), _u(N), _f(T_thresh), _p(gt), bg,
                                                   _p(grad_loss), _p(gt_depth), _f(depth_weight), _p(weights_sum), _p(depth), _p(image),
                                                   _p(grad_sigmas), _p(grad_rgbs), _p(grad_image), _p(grad_weights_sum), _p(loss),
                                                   _p(workspace), _stream()), "composite_rays_train_loss")

    zero_fills_march_rays = True  # march_rays(zero_unfilled=True): the kernel writes the zeros of the unfilled slots itself

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H, grid,
                   nears, fars, xyzs, dirs, deltas, noises, n_alive_dev=None, n_rows_out=None, zero_unfilled=False):
        """n_alive_dev / n_rows_out (build extension): device-side alive count and the sample-row count it implies;
        zero_unfilled: xyzs / dirs / deltas arrive uninitialised, the kernel zeroes what it does not fill; noises may be None"""
        _need(rays_o, torch.float32, "rays_o")
        _check(lib().s3d_march_rays(_u(n_alive), _u(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d),
                                    _f(bound), _f(dt_gamma), _u(max_steps), _u(Cc), _u(H), _p(grid), _p(nears),
                                    _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(noises), _nv(n_alive_dev), _nv(n_rows_out),
                                    _u(xyzs.shape[0]), C.c_int(int(zero_unfilled)), _stream()), "march_rays")

    @staticmethod
    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth,
                       image, n_alive_dev=None):
        for t, n in ((image, "image"), (deltas, "deltas"), (weights_sum, "weights_sum"), (depth, "depth"), (rays_t, "rays_t")):
            _need(t, torch.float32, n)
        _check(lib().s3d_composite_rays(_u(n_alive), _u(n_step), _f(T_thresh), _p(rays_alive), _p(rays_t),
                                        _p(sigmas), _p(rgbs), _p(deltas), _p(weights_sum), _p(depth), _p(image),
                                        _nv(n_alive_dev), C.c_int(_dt(sigmas)), C.c_int(_dt(rgbs)), _stream()), "composite_rays")

    # kernel choice handed to the library with every call (`path` arguments of seal3d_hip.h); binding-side state for
    # tests / experiments — the library itself keeps no process-wide switches
    _march_path = 0
    _composite_path = 0

    @staticmethod
    def set_composite_path(path):
        """0 = wave-per-ray compositing, 1 = lane-per-ray (tests / experiments)"""
        RaymarchingBackend._composite_path = int(path)

    @staticmethod
    def set_march_path(path):
        """0 = auto, 1 = lane-per-ray kernels, 2 = wave-per-ray kernels, 3 = wave-per-ray without the single-cascade fast path (tests / experiments)"""
        RaymarchingBackend._march_path = int(path)

    # --- build extension (not in the reference's native surface) ---
    @staticmethod
    def compact_alive(rays_alive, n, out, n_out, n_in_dev=None):
        nbytes = lib().s3d_compact_alive_workspace_size(_u(n))
        ws = _ws.get(nbytes, rays_alive.device)
        _check(lib().s3d_compact_alive(_p(rays_alive), _u(n), _p(out), _p(n_out), _p(ws), C.c_size_t(ws.numel()),
                                       _nv(n_in_dev), _stream()), "compact_alive")


_level_rows_cache = {}  # id(tensor) -> (weakref, version, rows)


def _max_level_rows(offsets):
    """max rows of one level.  `offsets` is the module's registered buffer (built on the host, grid.py:104-112); the
    one-time read-back happens on the first backward, before any graph capture, and is cached per tensor object."""
    hit = _level_rows_cache.get(id(offsets))
    if hit is not None and hit[0]() is offsets and hit[1] == offsets._version:
        return hit[2]
    if torch.cuda.is_current_stream_capturing():
        return 0  # unknown: the library falls back to direct atomics
    o = offsets.detach().cpu()
    v = int((o[1:] - o[:-1]).max().item()) if o.numel() > 1 else 0
    if len(_level_rows_cache) > 256:
        for k in [k for k, h in _level_rows_cache.items() if h[0]() is None]:
            del _level_rows_cache[k]
    _level_rows_cache[id(offsets)] = (weakref.ref(offsets), offsets._version, v)
    return v


class GridBackend:
    """gridencoder/src/gridencoder.h:12-15"""

    supports_bound = True  # forward/backward accept raw coordinates + `bound` (normalisation fused into the kernels)

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, dy_dx, gridtype, align_corners,
                            interp, bound=0.0, n_valid=None, live=None):
        _need(inputs, torch.float32, "inputs")
        _need(offsets, torch.int32, "offsets")
        if outputs.dtype != embeddings.dtype:
            raise RuntimeError("outputs must have the dtype of embeddings")
        _check(lib().s3d_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), _u(B), _u(D),
                                             _u(Cc), _u(L), _f(S), _u(H), _p(dy_dx), _u(gridtype),
                                             C.c_int(int(align_corners)), _u(interp), C.c_int(_dt(embeddings)),
                                             _f(bound), _nv(n_valid), *_live(live, B), _stream()), "grid_encode_forward")

    @staticmethod
    def grid_encode_forward_pair(inputs, emb_a, emb_b, offsets, out_a, out_b, B, D, Cc, L, S, H, gridtype, align_corners,
                                 interp, bound=0.0, n_valid=None, live=None):
        """two tables of one geometry on the same points in one launch (seal3d_hip.h: s3d_grid_encode_forward_pair)"""
        _need(inputs, torch.float32, "inputs")
        _need(offsets, torch.int32, "offsets")
        if not (out_a.dtype == emb_a.dtype == emb_b.dtype == out_b.dtype) or emb_a.shape != emb_b.shape:
            raise RuntimeError("grid_encode_forward_pair: both tables and both outputs share dtype and shape")
        _check(lib().s3d_grid_encode_forward_pair(_p(inputs), _p(emb_a), _p(emb_b), _p(offsets), _p(out_a), _p(out_b), _u(B), _u(D),
                                                  _u(Cc), _u(L), _f(S), _u(H), _u(gridtype), C.c_int(int(align_corners)),
                                                  _u(interp), C.c_int(_dt(emb_a)), _f(bound), _nv(n_valid), *_live(live, B),
                                                  _stream()), "grid_encode_forward_pair")

    @staticmethod
    def grid_corner_indices(inputs, offsets, corner_idx, B, D, Cc, L, S, H, gridtype, align_corners):
        _check(lib().s3d_grid_corner_indices(_p(inputs), _p(offsets), _p(corner_idx), _u(B), _u(D), _u(Cc), _u(L),
                                             _f(S), _u(H), _u(gridtype), C.c_int(int(align_corners)), _stream()),
               "grid_corner_indices")

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, dy_dx,
                             grad_inputs, gridtype, align_corners, interp, bound=0.0, n_valid=None, found_inf=None):
        _need(inputs, torch.float32, "inputs")
        if found_inf is not None:
            _need(found_inf, torch.float32, "found_inf")
        if grad_embeddings.dtype != grad.dtype:
            raise RuntimeError("grad_embeddings must have the dtype of grad")
        mlr = _max_level_rows(offsets)
        ws = _ws.get(lib().s3d_grid_encode_backward_workspace_size(_u(B), _u(D), _u(Cc), _u(L), _u(mlr),
                                                                   C.c_int(_dt(grad))), grad.device)
        ctl = _ctl.get(lib().s3d_grid_encode_backward_control_size(_u(D), _u(Cc), _u(L), _u(mlr), C.c_int(_dt(grad))),
                       grad.device) if B >= 8192 else None
        _check(lib().s3d_grid_encode_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets),
                                              _p(grad_embeddings), _u(mlr), _u(B), _u(D),
                                              _u(Cc), _u(L), _f(S), _u(H), _p(dy_dx), _p(grad_inputs), _u(gridtype),
                                              C.c_int(int(align_corners)), _u(interp), C.c_int(_dt(grad)), _p(ws),
                                              C.c_size_t(ws.numel()), _f(bound), _nv(n_valid),
                                              C.c_int(GridBackend._backward_path), _p(found_inf), _p(ctl),
                                              C.c_size_t(ctl.numel() if ctl is not None else 0), _stream()),
               "grid_encode_backward")

    class _GridAdam(C.Structure):
        _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("param_half", C.c_void_p),
                    ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                    ("step", C.c_void_p), ("grad_scale", C.c_void_p), ("lr_scale", C.c_void_p)]

    @staticmethod
    def grid_encode_backward_adam(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, gridtype, align_corners,
                                  interp, adam, bound=0.0, n_valid=None, found_inf=None):
        """backward of a table with its Adam update inside the accumulate kernel (include/seal3d_hip.h:
        s3d_grid_encode_backward_adam).  `adam`: dict(param, exp_avg, exp_avg_sq, param_half, lr, betas, eps, step, grad_scale,
        lr_scale).  Returns True when the update was applied there, False when the plain backward ran (the gradient is in
        `grad_embeddings`)."""
        _need(inputs, torch.float32, "inputs")
        if found_inf is not None:
            _need(found_inf, torch.float32, "found_inf")
        if grad_embeddings.dtype != grad.dtype:
            raise RuntimeError("grad_embeddings must have the dtype of grad")
        for k in ("param", "exp_avg", "exp_avg_sq"):
            _need(adam[k], torch.float32, k)
            if adam[k].shape != embeddings.shape or not adam[k].is_contiguous():
                raise RuntimeError(f"adam[{k!r}] must be a contiguous fp32 tensor of the table's shape")
        mlr = _max_level_rows(offsets)
        ws = _ws.get(lib().s3d_grid_encode_backward_workspace_size(_u(B), _u(D), _u(Cc), _u(L), _u(mlr), C.c_int(_dt(grad))), grad.device)
        ctl = _ctl.get(lib().s3d_grid_encode_backward_control_size(_u(D), _u(Cc), _u(L), _u(mlr), C.c_int(_dt(grad))),
                       grad.device) if B >= 8192 else None
        ga = GridBackend._GridAdam()
        ga.param, ga.exp_avg, ga.exp_avg_sq = adam["param"].data_ptr(), adam["exp_avg"].data_ptr(), adam["exp_avg_sq"].data_ptr()
        ga.param_half = adam["param_half"].data_ptr() if adam.get("param_half") is not None else None
        ga.lr, (ga.beta1, ga.beta2), ga.eps = float(adam["lr"]), adam["betas"], float(adam["eps"])
        ga.step = adam["step"].data_ptr()
        ga.grad_scale = adam["grad_scale"].data_ptr() if adam.get("grad_scale") is not None else None
        ga.lr_scale = adam["lr_scale"].data_ptr() if adam.get("lr_scale") is not None else None
        applied = C.c_int(0)
        _check(lib().s3d_grid_encode_backward_adam(_p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), _u(mlr), _u(B),
                                                   _u(D), _u(Cc), _u(L), _f(S), _u(H), _u(gridtype), C.c_int(int(align_corners)), _u(interp),
                                                   C.c_int(_dt(grad)), _p(ws), C.c_size_t(ws.numel()), _f(bound), _nv(n_valid),
                                                   _p(found_inf), _p(ctl), C.c_size_t(ctl.numel() if ctl is not None else 0),
                                                   C.byref(ga), C.byref(applied), _stream()), "grid_encode_backward_adam")
        return bool(applied.value)

    _backward_path = 0  # `path` argument of s3d_grid_encode_backward (binding-side state for tests / experiments)

    @staticmethod
    def set_backward_path(path):
        """0 = auto, 1 = direct global atomics, 2 = binned: partition + LDS accumulate (tests / experiments)"""
        GridBackend._backward_path = int(path)

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, Cc, L, S, H, gridtype, align_corners):
        _need(embeddings, torch.float32, "embeddings")
        _need(inputs, torch.float32, "inputs")
        _check(lib().s3d_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), _f(weight), _u(B),
                                              _u(D), _u(Cc), _u(L), _f(S), _u(H), _u(gridtype),
                                              C.c_int(int(align_corners)), _stream()), "grad_total_variation")


class SHBackend:
    """shencoder/src/shencoder.h:9-10"""

    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, Cc, dy_dx):
        _need(inputs, torch.float32, "inputs")
        _check(lib().s3d_sh_encode_forward(_p(inputs), _p(outputs), _u(B), _u(D), _u(Cc), _p(dy_dx), _stream()),
               "sh_encode_forward")

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, Cc, dy_dx, grad_inputs):
        _need(grad, torch.float32, "grad")
        _check(lib().s3d_sh_encode_backward(_p(grad), _p(inputs), _u(B), _u(D), _u(Cc), _p(dy_dx), _p(grad_inputs),
                                            _stream()), "sh_encode_backward")


class FreqBackend:
    """freqencoder/src/freqencoder.h:7,10"""

    @staticmethod
    def freq_encode_forward(inputs, B, D, deg, Cc, outputs):
        _need(inputs, torch.float32, "inputs")
        _check(lib().s3d_freq_encode_forward(_p(inputs), _u(B), _u(D), _u(deg), _u(Cc), _p(outputs), _stream()),
               "freq_encode_forward")

    @staticmethod
    def freq_encode_pack_forward(a, d, deg1, deg2, out, n_valid=None):
        """out fp16 [B, ld] = [freq(a) | freq(d) | 0]: a fp16 [B, D1], d fp32 [B, D2] (seal3d_hip.h)"""
        _need(a, torch.float16, "a"); _need(d, torch.float32, "d"); _need(out, torch.float16, "out")
        B = a.shape[0]
        if not (a.is_contiguous() and d.is_contiguous() and out.is_contiguous()) or d.shape[0] != B or out.shape[0] != B:
            raise RuntimeError("freq_encode_pack_forward: contiguous a [B, D1], d [B, D2], out [B, ld]")
        _check(lib().s3d_freq_encode_pack_forward(_p(a), _p(d), _u(B), _u(a.shape[1]), _u(deg1), _u(d.shape[1]), _u(deg2), _u(out.shape[1]),
                                                  _p(out), _nv(n_valid), _stream()), "freq_encode_pack_forward")

    @staticmethod
    def freq_encode_pack_backward(grad, a, deg1, grad_a, n_valid=None):
        """grad_a fp16 [B, ldg] (columns behind D1 zero) from the packed row's gradient fp16 [B, ld]"""
        _need(grad, torch.float16, "grad"); _need(a, torch.float16, "a"); _need(grad_a, torch.float16, "grad_a")
        B = a.shape[0]
        if not (grad.is_contiguous() and a.is_contiguous() and grad_a.is_contiguous()) or grad.shape[0] != B or grad_a.shape[0] != B:
            raise RuntimeError("freq_encode_pack_backward: contiguous grad [B, ld], a [B, D1], grad_a [B, ldg]")
        _check(lib().s3d_freq_encode_pack_backward(_p(grad), _p(a), _u(B), _u(a.shape[1]), _u(deg1), _u(grad.shape[1]), _u(grad_a.shape[1]),
                                                   _p(grad_a), _nv(n_valid), _stream()), "freq_encode_pack_backward")

    @staticmethod
    def freq_encode_backward(grad, outputs, B, D, deg, Cc, grad_inputs):
        _need(grad, torch.float32, "grad")
        _check(lib().s3d_freq_encode_backward(_p(grad), _p(outputs), _u(B), _u(D), _u(deg), _u(Cc), _p(grad_inputs),
                                              _stream()), "freq_encode_backward")


class FFMLPBackend:
    """ffmlp/src/ffmlp.h:8-14"""

    @staticmethod
    def allocate_splitk(n):
        _check(lib().s3d_ffmlp_allocate_splitk(C.c_size_t(int(n))), "allocate_splitk")

    @staticmethod
    def free_splitk():
        _check(lib().s3d_ffmlp_free_splitk(), "free_splitk")

    @staticmethod
    def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                      output_activation, forward_buffer, outputs, input_layout=0, n_valid=None, rgb_head=None, mid=None):
        """`mid` = (dirs f32 [B,3], sigma f32 [B], color_in f16 [B,32], h0 f16 [B]): the density head (seal3d_hip.h)"""
        _need(inputs, torch.float16, "inputs")
        _need(weights, torch.float16, "weights")
        if rgb_head is not None:
            _need(rgb_head, torch.float32, "rgb_head")
        _check(lib().s3d_ffmlp_forward(_p(inputs), _p(weights), _u(B), _u(input_dim), _u(output_dim), _u(hidden_dim),
                                       _u(num_layers), _u(activation), _u(output_activation), _p(forward_buffer),
                                       _p(outputs), C.c_int(int(input_layout)), _nv(n_valid), _p(rgb_head), *_mid_fwd_args(mid, B),
                                       _stream()), "ffmlp_forward")

    @staticmethod
    def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                        output_activation, inference_buffer, outputs, input_layout=0, n_valid=None, rgb_head=None, mid=None):
        _need(inputs, torch.float16, "inputs")
        _need(weights, torch.float16, "weights")
        if rgb_head is not None:
            _need(rgb_head, torch.float32, "rgb_head")
        _check(lib().s3d_ffmlp_inference(_p(inputs), _p(weights), _u(B), _u(input_dim), _u(output_dim),
                                         _u(hidden_dim), _u(num_layers), _u(activation), _u(output_activation),
                                         _p(inference_buffer), _p(outputs), C.c_int(int(input_layout)), _nv(n_valid),
                                         _p(rgb_head), *_mid_fwd_args(mid, B), _stream()), "ffmlp_inference")

    @staticmethod
    def ngp_pair_inference(inputs, weights_sigma, weights_color, B, hidden_dim, num_layers_sigma, num_layers_color, dirs,
                           sigma, rgb, input_layout=0, n_valid=None, color_in=None, h0=None, enc_color=None):
        """density network + head + colour network + sigmoid in one launch (seal3d_hip.h: s3d_ffmlp_ngp_pair_inference)"""
        for t, n in ((inputs, "inputs"), (weights_sigma, "weights_sigma"), (weights_color, "weights_color")):
            _need(t, torch.float16, n)
        for t, n in ((dirs, "dirs"), (sigma, "sigma"), (rgb, "rgb")):
            _need(t, torch.float32, n)
        _check(lib().s3d_ffmlp_ngp_pair_inference(_p(inputs), _p(weights_sigma), _p(weights_color), _u(B), _u(hidden_dim),
                                                  _u(num_layers_sigma), _u(num_layers_color), C.c_int(int(input_layout)),
                                                  _nv(n_valid), _p(dirs), _p(sigma), _p(rgb), _p(color_in), _p(h0), _p(enc_color),
                                                  _stream()),
               "ffmlp_ngp_pair_inference")

    @staticmethod
    def fused_backward_supported(input_dim, output_dim, hidden_dim, num_layers, activation):
        """True when ffmlp_backward can run without forward_buffer / backward_buffer (re-computing fused kernel)"""
        return bool(lib().s3d_ffmlp_fused_backward_supported(_u(input_dim), _u(output_dim), _u(hidden_dim),
                                                             _u(num_layers), _u(activation)))

    @staticmethod
    def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers,
                       activation, output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights,
                       input_layout=0, accumulate=False, n_valid=None, found_inf=None, grad_rgb=None, rgb_head=None, mid=None,
                       workspace=None, defer_reduce=False):
        """`mid` = (grad_sigma f32 [B] or None, grad_color_in f16 [B,32], h0 f16 [B]): the density head's gradients instead of
        `grad` (seal3d_hip.h); `defer_reduce` + `workspace` (a caller-owned uint8 tensor of ffmlp_backward_workspace_size bytes):
        the weight gradient stays as partial sums in it for wgrad_reduce_pair"""
        if grad_rgb is not None:
            _need(grad_rgb, torch.float32, "grad_rgb"); _need(rgb_head, torch.float32, "rgb_head")
        elif mid is None:
            _need(grad, torch.float16, "grad")
        if found_inf is not None:
            _need(found_inf, torch.float32, "found_inf")
        nbytes = lib().s3d_ffmlp_backward_workspace_size(_u(input_dim), _u(output_dim), _u(hidden_dim),
                                                        _u(num_layers))
        ws = workspace if workspace is not None else _ws.get(nbytes, inputs.device)
        if ws.numel() * ws.element_size() < nbytes:
            raise RuntimeError("ffmlp_backward: workspace too small")
        _check(lib().s3d_ffmlp_backward(_p(grad), _p(inputs), _p(weights), _p(forward_buffer), _u(B), _u(input_dim),
                                        _u(output_dim), _u(hidden_dim), _u(num_layers), _u(activation),
                                        _u(output_activation), C.c_int(int(bool(calc_grad_inputs))),
                                        _p(backward_buffer), _p(grad_inputs if calc_grad_inputs else None),
                                        _p(grad_weights), _p(ws), C.c_size_t(ws.numel()), C.c_int(int(input_layout)),
                                        C.c_int(2 if defer_reduce else int(bool(accumulate))), _nv(n_valid), _p(found_inf), _p(grad_rgb), _p(rgb_head),
                                        *_mid_bwd_args(mid, B), _stream()), "ffmlp_backward")

    @staticmethod
    def backward_workspace_bytes(input_dim, output_dim, hidden_dim, num_layers):
        return int(lib().s3d_ffmlp_backward_workspace_size(_u(input_dim), _u(output_dim), _u(hidden_dim), _u(num_layers)))

    @staticmethod
    def wgrad_reduce_pair(a, b):
        """finish two deferred backward calls: a, b = (workspace, B, input_dim, hidden_dim, num_layers, grad_weights, accumulate, found_inf)"""
        (ws_a, B_a, in_a, hid_a, nl_a, gw_a, acc_a, fi_a), (ws_b, B_b, in_b, hid_b, nl_b, gw_b, acc_b, fi_b) = a, b
        _need(gw_a, torch.float16, "grad_weights"); _need(gw_b, torch.float16, "grad_weights")
        _check(lib().s3d_ffmlp_wgrad_reduce_pair(_p(ws_a), _u(B_a), _u(in_a), _u(hid_a), _u(nl_a), _p(gw_a), C.c_int(int(bool(acc_a))),
                                                 _p(fi_a), _p(ws_b), _u(B_b), _u(in_b), _u(hid_b), _u(nl_b), _p(gw_b),
                                                 C.c_int(int(bool(acc_b))), _p(fi_b), _stream()), "ffmlp_wgrad_reduce_pair")


def _mid_fwd_args(mid, B):
    """the four density-head arguments of s3d_ffmlp_forward / _inference from (dirs, sigma, color_in, h0) or None"""
    if mid is None:
        return (C.c_void_p(0),) * 4
    dirs, sigma, cin, h0 = mid
    _need(dirs, torch.float32, "mid dirs"); _need(sigma, torch.float32, "mid sigma")
    _need(cin, torch.float16, "mid color_in"); _need(h0, torch.float16, "mid h0")
    if dirs.numel() != 3 * B or sigma.numel() != B or cin.numel() != 32 * B or h0.numel() != B:
        raise RuntimeError("ffmlp density head: dirs [B,3], sigma [B], color_in [B,32], h0 [B]")
    return _p(dirs), _p(sigma), _p(cin), _p(h0)


def _mid_bwd_args(mid, B):
    """the three density-head arguments of s3d_ffmlp_backward from (grad_sigma or None, grad_color_in, h0) or None"""
    if mid is None:
        return (C.c_void_p(0),) * 3
    if mid[0] is not None:
        _need(mid[0], torch.float32, "mid grad_sigma")
    _need(mid[1], torch.float16, "mid grad_color_in"); _need(mid[2], torch.float16, "mid h0")
    if mid[1].numel() != 32 * B or mid[2].numel() != B or (mid[0] is not None and mid[0].numel() != B):
        raise RuntimeError("ffmlp density head: grad_sigma [B], grad_color_in [B,32], h0 [B]")
    return _p(mid[0]), _p(mid[1]), _p(mid[2])


class _AdamTensor(C.Structure):
    """seal3d_hip.h: s3d_adam_tensor"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("param_half", C.c_void_p), ("n", C.c_size_t), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("grad_dtype", C.c_int), ("consume", C.c_int), ("pack_cols", C.c_uint32),
                ("pack_stride", C.c_uint32), ("l1", C.c_float)]


class OptimBackend:
    """csrc/optim.hip — Adam + GradScaler bookkeeping straight from the (fp16) gradients"""

    @staticmethod
    def grads_nonfinite(grad, found_inf):
        _need(found_inf, torch.float32, "found_inf")
        _check(lib().s3d_grads_nonfinite(_p(grad), C.c_size_t(grad.numel()), C.c_int(_dt(grad)), _p(found_inf), _stream()),
               "grads_nonfinite")

    @staticmethod
    def adam_step(param, grad, exp_avg, exp_avg_sq, param_half, lr, beta1, beta2, eps, step, grad_scale, found_inf, lr_scale=None):
        _need(param, torch.float32, "param")
        _need(exp_avg, torch.float32, "exp_avg")
        _need(exp_avg_sq, torch.float32, "exp_avg_sq")
        if grad.numel() != param.numel() or not grad.is_contiguous() or not param.is_contiguous():
            raise RuntimeError("adam_step: param and grad must be contiguous and of equal size")
        if param_half is not None:
            _need(param_half, torch.float16, "param_half")
        _check(lib().s3d_adam_step(_p(param), _p(grad), C.c_int(_dt(grad)), _p(exp_avg), _p(exp_avg_sq), _p(param_half),
                                   C.c_size_t(param.numel()), _f(lr), _f(beta1), _f(beta2), _f(eps), _p(step),
                                   _p(grad_scale), _p(found_inf), _p(lr_scale), _stream()), "adam_step")

    @staticmethod
    def adam_step_multi(items, step, grad_scale, found_inf, consume_grads=False, lr_scale=None):
        """`items`: (param, grad, exp_avg, exp_avg_sq, param_half or None, lr, beta1, beta2, eps[, consume[, l1]]) per tensor —
        adam_step for all of them in one launch; `consume_grads` (all tensors) / the optional tenth element (that tensor): the
        gradient is cleared behind the read; eleventh element: coefficient of an L1 penalty whose gradient the update adds
        (seal3d_hip.h)"""
        arr = (_AdamTensor * len(items))()
        for a, item in zip(arr, items):
            param, grad, exp_avg, exp_avg_sq, param_half, lr, beta1, beta2, eps = item[:9]
            a.consume = int(bool(item[9])) if len(item) > 9 else 0
            a.l1 = float(item[10]) if len(item) > 10 else 0.0
            _need(param, torch.float32, "param"); _need(exp_avg, torch.float32, "exp_avg"); _need(exp_avg_sq, torch.float32, "exp_avg_sq")
            packed = grad.dim() == 2 and not grad.is_contiguous()
            if packed:
                # [rows, cols] views into a packed weight buffer (row stride > cols): gradient and fp16 copy share the layout
                if (tuple(grad.shape) != tuple(param.shape) or grad.stride(1) != 1 or not grad.is_cuda or not param.is_contiguous()
                        or (param_half is not None and (param_half.stride() != grad.stride() or param_half.dtype != torch.float16
                                                        or not param_half.is_cuda))):
                    raise RuntimeError("adam_step_multi: a packed gradient is a [rows, cols] row-strided view; the fp16 copy shares its strides")
                a.pack_cols, a.pack_stride = int(grad.shape[1]), int(grad.stride(0))
                a.param, a.grad, a.exp_avg, a.exp_avg_sq = _p(param), C.c_void_p(grad.data_ptr()), _p(exp_avg), _p(exp_avg_sq)
                a.param_half = C.c_void_p(param_half.data_ptr() if param_half is not None else 0)
            else:
                if grad.numel() != param.numel() or not grad.is_contiguous() or not param.is_contiguous():
                    raise RuntimeError("adam_step_multi: param and grad must be contiguous and of equal size")
                if param_half is not None:
                    _need(param_half, torch.float16, "param_half")
                a.param, a.grad, a.exp_avg, a.exp_avg_sq = _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq)
                a.param_half = _p(param_half)
            a.n = param.numel()
            a.lr, a.beta1, a.beta2, a.eps = float(lr), float(beta1), float(beta2), float(eps)
            a.grad_dtype = _dt(grad)
        _check(lib().s3d_adam_step_multi(arr, C.c_int32(len(items)), _p(step), _p(grad_scale), _p(found_inf), _p(lr_scale),
                                         C.c_int(int(bool(consume_grads))), _stream()), "adam_step_multi")

    @staticmethod
    def scaler_update(scale, growth_tracker, found_inf, growth_factor, backoff_factor, growth_interval, adam_step=None):
        _need(scale, torch.float32, "scale"); _need(growth_tracker, torch.int32, "growth_tracker")
        if adam_step is not None:
            _need(adam_step, torch.float32, "adam_step")
        _check(lib().s3d_scaler_update(_p(scale), _p(growth_tracker), _p(found_inf), _f(growth_factor), _f(backoff_factor),
                                       C.c_int32(int(growth_interval)), _p(adam_step), _stream()), "scaler_update")

    @staticmethod
    def adam_advance(step, found_inf):
        _check(lib().s3d_adam_advance(_p(step), _p(found_inf), _stream()), "adam_advance")

    @staticmethod
    def step_epilogue(scale, growth_tracker, found_inf, growth_factor, backoff_factor, growth_interval, adam_step, loss, counter,
                      loss_ring, counter_ring, cursor):
        """scaler_update + step_ring_push in one launch (seal3d_hip.h)"""
        _need(scale, torch.float32, "scale"); _need(growth_tracker, torch.int32, "growth_tracker")
        _need(counter, torch.int32, "counter"); _need(counter_ring, torch.int32, "counter_ring"); _need(cursor, torch.int32, "cursor")
        if adam_step is not None:
            _need(adam_step, torch.float32, "adam_step")
        if loss is not None:
            _need(loss, torch.float32, "loss"); _need(loss_ring, torch.float32, "loss_ring")
        ring = counter_ring.shape[0]
        if cursor.numel() < 2 or counter_ring.numel() != 2 * ring or (loss is not None and loss_ring.numel() < 1):
            raise RuntimeError("step_epilogue: cursor is int32[2], rings are [ring, 2] / [loss_slots]")
        slots = 0 if loss is None or loss_ring.numel() == ring else loss_ring.numel()
        _check(lib().s3d_step_epilogue(_p(scale), _p(growth_tracker), _p(found_inf), _f(growth_factor), _f(backoff_factor),
                                       C.c_int32(int(growth_interval)), _p(adam_step), _p(loss),
                                       _p(counter), _p(loss_ring if loss is not None else None), _p(counter_ring), _p(cursor),
                                       C.c_int32(ring), C.c_int32(slots), _stream()), "step_epilogue")

    @staticmethod
    def step_ring_push(loss, counter, loss_ring, counter_ring, cursor):
        """file `loss` [] / `counter` [2] in slot *cursor of the rings, clear the counter, advance the cursor (seal3d_hip.h)"""
        _need(counter, torch.int32, "counter"); _need(counter_ring, torch.int32, "counter_ring")
        _need(cursor, torch.int32, "cursor")
        if cursor.numel() < 2:
            raise RuntimeError("step_ring_push: cursor is int32[2] = {slot, running step number}")
        if loss is not None:
            _need(loss, torch.float32, "loss"); _need(loss_ring, torch.float32, "loss_ring")
        ring = counter_ring.shape[0]
        if not counter_ring.is_contiguous() or counter_ring.numel() != 2 * ring or (loss is not None and loss_ring.numel() < 1):
            raise RuntimeError("step_ring_push: rings must be contiguous [ring, 2] / [loss_slots]")
        slots = 0 if loss is None or loss_ring.numel() == ring else loss_ring.numel()
        _check(lib().s3d_step_ring_push(_p(loss), _p(counter), _p(loss_ring if loss is not None else None), _p(counter_ring),
                                        _p(cursor), C.c_int32(ring), C.c_int32(slots), _stream()), "step_ring_push")


class NgpHeadBackend:
    """csrc/ngp_head.hip — the elementwise glue between the two MLPs of nerf/network_ff.py"""

    @staticmethod
    def mid_forward(h, dirs, sigma, color_in, n_valid=None):
        _need(h, torch.float16, "h"); _need(dirs, torch.float32, "dirs")
        _need(sigma, torch.float32, "sigma"); _need(color_in, torch.float16, "color_in")
        _check(lib().s3d_ngp_mid_forward(_p(h), _p(dirs), _u(h.shape[0]), _p(sigma), _p(color_in), _nv(n_valid), _stream()),
               "ngp_mid_forward")

    @staticmethod
    def mid_backward(grad_color_in, grad_sigma, h, grad_h, n_valid=None):
        _need(grad_color_in, torch.float16, "grad_color_in"); _need(grad_h, torch.float16, "grad_h")
        if grad_sigma is not None:
            _need(grad_sigma, torch.float32, "grad_sigma")
        _check(lib().s3d_ngp_mid_backward(_p(grad_color_in), _p(grad_sigma), _p(h), _u(h.shape[0]), _p(grad_h), _nv(n_valid),
                                          _stream()), "ngp_mid_backward")

    @staticmethod
    def mid2_forward(h, dirs, enc_color, sigma, color_in, n_valid=None):
        """two-encoder network: color_in [B,64] = [SH | geo | enc_color | 0]; enc_color level-major [16,B,2] fp16"""
        _need(h, torch.float16, "h"); _need(dirs, torch.float32, "dirs"); _need(enc_color, torch.float16, "enc_color")
        _need(sigma, torch.float32, "sigma"); _need(color_in, torch.float16, "color_in")
        B = h.shape[0]
        if tuple(enc_color.shape) != (16, B, 2) or tuple(color_in.shape) != (B, 64):
            raise RuntimeError("mid2_forward: enc_color must be [16,B,2], color_in [B,64]")
        _check(lib().s3d_ngp_mid2_forward(_p(h), _p(dirs), _p(enc_color), _u(B), _p(sigma), _p(color_in), _nv(n_valid), _stream()),
               "ngp_mid2_forward")

    @staticmethod
    def mid2_backward(grad_color_in, grad_sigma, h, grad_h, grad_enc_color, n_valid=None):
        _need(grad_color_in, torch.float16, "grad_color_in"); _need(grad_h, torch.float16, "grad_h")
        if grad_sigma is not None:
            _need(grad_sigma, torch.float32, "grad_sigma")
        if grad_enc_color is not None:
            _need(grad_enc_color, torch.float16, "grad_enc_color")
        # h: the density network's [B, 16] output, or its first column [B] alone (the one-launch pair keeps only that)
        _need(h, torch.float16, "h")
        _check(lib().s3d_ngp_mid2_backward(_p(grad_color_in), _p(grad_sigma), _p(h), _u(16 if h.dim() == 2 else 1), _u(h.shape[0]),
                                           _p(grad_h), _p(grad_enc_color), _nv(n_valid), _stream()), "ngp_mid2_backward")

    @staticmethod
    def rgb_forward(out, rgb, n_valid=None):
        _need(out, torch.float16, "out"); _need(rgb, torch.float32, "rgb")
        _check(lib().s3d_ngp_rgb_forward(_p(out), _u(out.shape[0]), _p(rgb), _nv(n_valid), _stream()), "ngp_rgb_forward")

    @staticmethod
    def rgb_backward(grad_rgb, rgb, grad_out, n_valid=None):
        _need(grad_rgb, torch.float32, "grad_rgb"); _need(rgb, torch.float32, "rgb"); _need(grad_out, torch.float16, "grad_out")
        _check(lib().s3d_ngp_rgb_backward(_p(grad_rgb), _p(rgb), _u(rgb.shape[0]), _p(grad_out), _nv(n_valid), _stream()),
               "ngp_rgb_backward")

    @staticmethod
    def bg_mse_forward(image, weights_sum, gt, bg_rgb, loss, grad_loss=None, grad_image=None, grad_weights_sum=None,
                       depth=None, gt_depth=None, depth_weight=1.0):
        """`depth`, `gt_depth` (both or neither): + depth_weight * L1(nan_to_num(depth), gt_depth) in the loss VALUE (seal3d_hip.h)"""
        for t, n in ((image, "image"), (weights_sum, "weights_sum"), (gt, "gt"), (loss, "loss")):
            _need(t, torch.float32, n)
        if grad_loss is not None:
            for t, n in ((grad_loss, "grad_loss"), (grad_image, "grad_image"), (grad_weights_sum, "grad_weights_sum")):
                _need(t, torch.float32, n)
        if depth is not None:
            _need(depth, torch.float32, "depth"); _need(gt_depth, torch.float32, "gt_depth")
            if depth.numel() != image.shape[0] or gt_depth.numel() != image.shape[0]:
                raise RuntimeError("bg_mse_forward: depth and gt_depth hold one value per ray")
        bg = (C.c_float * 3)(*[float(v) for v in bg_rgb])
        _check(lib().s3d_bg_mse_forward(_p(image), _p(weights_sum), _p(gt), bg, _u(image.shape[0]), _p(loss), _p(grad_loss),
                                        _p(grad_image), _p(grad_weights_sum), _p(depth), _p(gt_depth), _f(depth_weight), _stream()),
               "bg_mse_forward")

    @staticmethod
    def bg_targets(image, weights_sum, depth, bg_rgb, out_rgb, out_depth=None):
        """out_rgb = nan_to_num(image + (1 - weights_sum) * bg), out_depth = nan_to_num(depth): a teacher render's targets"""
        for t, n in ((image, "image"), (weights_sum, "weights_sum"), (out_rgb, "out_rgb")):
            _need(t, torch.float32, n)
        N = image.shape[0]
        if out_rgb.numel() != 3 * N or weights_sum.numel() != N:
            raise RuntimeError("bg_targets: image / out_rgb [N,3], weights_sum [N]")
        if out_depth is not None:
            _need(depth, torch.float32, "depth"); _need(out_depth, torch.float32, "out_depth")
            if depth.numel() != N or out_depth.numel() != N:
                raise RuntimeError("bg_targets: depth / out_depth [N]")
        bg = (C.c_float * 3)(*[float(v) for v in bg_rgb])
        _check(lib().s3d_bg_targets(_p(image), _p(weights_sum), _p(depth if out_depth is not None else None), bg, _u(N),
                                    _p(out_rgb), _p(out_depth), _stream()), "bg_targets")

    _l1_ws = {}

    @staticmethod
    def l1_pair_loss(sigma, color, gt_sigma, gt_color, n_total, loss, grad_loss=None, grad_sigma=None, grad_color=None):
        """Seal-3D's pretraining loss L1(sigma) + L1(colour) (+ its gradients for a known upstream gradient), seal3d_hip.h"""
        for t, n in ((sigma, "sigma"), (color, "color"), (gt_sigma, "gt_sigma"), (gt_color, "gt_color"), (loss, "loss")):
            _need(t, torch.float32, n)
        n, n_rows = gt_sigma.numel(), sigma.numel()
        if color.numel() != 3 * n_rows or n_rows < n or gt_color.numel() != 3 * n:
            raise RuntimeError("l1_pair_loss: sigma [n_rows] / color [n_rows,3], gt_sigma [n] / gt_color [n,3], n_rows >= n")
        if grad_loss is not None:
            for t, nm in ((grad_loss, "grad_loss"), (grad_sigma, "grad_sigma"), (grad_color, "grad_color")):
                _need(t, torch.float32, nm)
        # one workspace (partial sums + the last-block ticket) per (device, stream): two launches in flight on different
        # streams — a pretraining graph on a side stream next to an eager call, two trainers — must not share a ticket
        key = (sigma.device.index, torch.cuda.current_stream(sigma.device).cuda_stream)
        ws = NgpHeadBackend._l1_ws.get(key)
        if ws is None:  # (zeroed once: the kernel leaves its ticket word zero)
            n_ws = lib().s3d_l1_pair_workspace_size() // 4
            if torch.cuda.is_current_stream_capturing():
                # (torch captures every graph on its own stream: a buffer first asked for during a capture is allocated from
                #  that graph's pool and zeroed by a fill the graph replays — it is not cached beyond the capture)
                ws = torch.zeros(n_ws, dtype=torch.float32, device=sigma.device)
            else:
                ws = NgpHeadBackend._l1_ws[key] = torch.zeros(n_ws, dtype=torch.float32, device=sigma.device)
        _check(lib().s3d_l1_pair_loss(_p(sigma), _p(color), _p(gt_sigma), _p(gt_color), _u(n), _u(n_rows), _u(n_total), _p(loss), _p(grad_loss),
                                      _p(grad_sigma), _p(grad_color), _p(ws), _stream()), "l1_pair_loss")

    @staticmethod
    def bg_mse_backward(image, weights_sum, gt, bg_rgb, grad_loss, grad_image, grad_weights_sum):
        _need(grad_loss, torch.float32, "grad_loss")
        bg = (C.c_float * 3)(*[float(v) for v in bg_rgb])
        _check(lib().s3d_bg_mse_backward(_p(image), _p(weights_sum), _p(gt), bg, _u(image.shape[0]), _p(grad_loss),
                                         _p(grad_image), _p(grad_weights_sum), _stream()), "bg_mse_backward")


class SealBackend:
    """csrc/seal.hip — Seal-3D's bbox proxy mapper (SealNeRF/seal_utils.py:132-279, 630-685) on the device"""

    @staticmethod
    def bbox_map(points, dirs, host, out_points, out_dirs, mask, n_valid=None):
        """`host`: dict of float32 numpy arrays (triangles, bounds, inv_transform, inv_rotation, inv_scale, center and
        optionally empty_bound, map_source) — the edit's constants live on the host."""
        import numpy as np
        _need(points, torch.float32, "points")
        if mask.dtype != torch.uint8:
            raise RuntimeError("mask must be uint8")

        def hp(name):
            a = host.get(name)
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=np.float32)
            return a, a.ctypes.data_as(C.POINTER(C.c_float))
        keep = [hp(k) for k in ("triangles", "bounds", "inv_transform", "inv_rotation", "inv_scale", "center", "empty_bound",
                                "map_source")]
        ptr = [k[1] for k in keep]
        _check(lib().s3d_seal_bbox_map(_p(points), _p(dirs), _u(points.shape[0]), ptr[0], _u(keep[0][0].shape[0]), ptr[1],
                                       _u(keep[1][0].shape[0]), ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], ptr[7], _p(out_points),
                                       _p(out_dirs), _p(mask), _nv(n_valid), _stream()), "seal_bbox_map")


    @staticmethod
    def map_color(rgbs, mask, hsv, rgb_target, light_offset, out, stats=None, n_valid=None):
        """colour edit of the moved samples (include/seal3d_hip.h: s3d_seal_map_color); `hsv` / `rgb_target`: 3 floats or None"""
        if rgbs.dtype not in (torch.float32, torch.float16) or out.dtype != rgbs.dtype or not rgbs.is_contiguous() or not out.is_contiguous():
            raise RuntimeError("map_color: contiguous f32 / f16 colours")
        if mask.dtype != torch.uint8:
            raise RuntimeError("mask must be uint8")
        h = (C.c_float * 3)(*[float(v) for v in hsv]) if hsv is not None else None
        t = (C.c_float * 3)(*[float(v) for v in rgb_target]) if rgb_target is not None else None
        if t is not None and stats is None:
            stats = torch.empty(2, dtype=torch.int64, device=rgbs.device)
        _check(lib().s3d_seal_map_color(_p(rgbs), _p(mask), _u(rgbs.shape[0]), C.c_int(_dt(rgbs)), h, t, C.c_float(float(light_offset)),
                                        _p(out), _p(stats), _nv(n_valid), _stream()), "seal_map_color")


def _zeros_like_many(tensors, words=0):
    """zero tensors shaped like `tensors` (fp32, contiguous) + `words` zero int32 words, carved out of ONE filled buffer: one fill
    launch instead of one per tensor (seven to eight ~5 us launches per factor backward otherwise); segments start on 16 bytes"""
    sizes = [(t.numel() + 3) // 4 * 4 for t in tensors]
    flat = torch.zeros(sum(sizes) + (words + 3) // 4 * 4, dtype=torch.float32, device=tensors[0].device)
    out, off = [], 0
    for t, n in zip(tensors, sizes):
        out.append(flat[off:off + t.numel()].view(t.shape))
        off += n
    return out, (flat[off:off + words].view(torch.int32) if words else None)


class VmBackend:
    """csrc/tensorf.hip — TensoRF vector-matrix features (tensoRF/network.py:112-153 of the reference)"""

    @staticmethod
    def features_forward(x, planes, lines, resolution, reduce, out, n_valid=None, shadows=None):
        """x [N,3] fp32; planes[i] [1,R_i,H,W] / lines[i] [1,R_i,D,1] fp32 (the reference's parameter shapes);
        out [N] (reduce) or [sum R_i, N]"""
        _need(x, torch.float32, "x"); _need(out, torch.float32, "out")
        for t in list(planes) + list(lines):
            _need(t, torch.float32, "factor")
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("vm features: factors must be contiguous GPU tensors")
        if not x.is_contiguous() or x.shape[-1] != 3:
            raise RuntimeError("vm features: x must be contiguous [N,3]")
        ptr3 = C.c_void_p * 3
        u3 = C.c_uint32 * 3
        pl = ptr3(*[t.data_ptr() for t in planes])
        ln = ptr3(*[t.data_ptr() for t in lines])
        rank = u3(*[int(t.shape[1]) for t in planes])
        res = u3(*[int(r) for r in resolution])
        _check(lib().s3d_vm_features_forward(_p(x), _u(x.shape[0]), pl, ln, rank, res, C.c_int(int(bool(reduce))), _p(out),
                                             *_shadow2(shadows), _nv(n_valid), _stream()), "vm_features_forward")

    @staticmethod
    def aabb_normalize(x, aabb, out):
        """out = 2 (x - aabb[:3]) / (aabb[3:] - aabb[:3]) - 1 per row of x [N, 3] (seal3d_hip.h)"""
        _need(x, torch.float32, "x"); _need(aabb, torch.float32, "aabb"); _need(out, torch.float32, "out")
        if not (x.is_contiguous() and out.is_contiguous() and aabb.is_contiguous()) or x.shape[-1] != 3 or aabb.numel() != 6 or out.shape != x.shape:
            raise RuntimeError("aabb_normalize: contiguous x / out [N, 3], aabb [6]")
        _check(lib().s3d_aabb_normalize(_p(x), _p(aabb), _u(x.shape[0]), _p(out), _stream()), "aabb_normalize")

    @staticmethod
    def weighted_abs_sum(tensors, weights, out):
        """out (fp32 scalar tensor) = sum_i weights[i] * sum |tensors[i]| (seal3d_hip.h)"""
        _need(out, torch.float32, "out")
        n = len(tensors)
        for t in tensors:
            _need(t, torch.float32, "tensor")
            if not t.is_contiguous() or not t.is_cuda:
                raise RuntimeError("weighted_abs_sum: contiguous GPU tensors")
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        numel = (C.c_uint64 * n)(*[t.numel() for t in tensors])
        ws = (C.c_float * n)(*[float(w) for w in weights])
        work = torch.empty(int(lib().s3d_weighted_abs_sum_workspace_size()) // 4, dtype=torch.float32, device=out.device)
        _check(lib().s3d_weighted_abs_sum(ptrs, numel, ws, C.c_int32(n), _p(out), _p(work), _stream()), "weighted_abs_sum")

    @staticmethod
    def _chain_args(mats, padded_rows, ld):
        n = len(mats)
        for m in mats:
            _need(m, torch.float32, "matrix")
            if m.dim() != 2 or not m.is_contiguous() or not m.is_cuda:
                raise RuntimeError("linear chain pack: contiguous fp32 GPU matrices")
        u = C.c_uint32 * n
        return ((C.c_void_p * n)(*[m.data_ptr() for m in mats]), u(*[m.shape[0] for m in mats]), u(*[m.shape[1] for m in mats]),
                u(*[int(r) for r in padded_rows]), u(*[int(v) for v in ld]), C.c_int32(n))

    @staticmethod
    def pack_linear_chain(mats, padded_rows, ld, flat):
        """flat fp16 = the matrices in the ffmlp layout, zero padded (seal3d_hip.h)"""
        _need(flat, torch.float16, "flat")
        if flat.numel() != sum(int(r) * int(v) for r, v in zip(padded_rows, ld)) or not flat.is_contiguous():
            raise RuntimeError("pack_linear_chain: flat holds sum(padded_rows * ld) elements")
        a = VmBackend._chain_args(mats, padded_rows, ld)
        _check(lib().s3d_pack_linear_chain(a[0], a[1], a[2], a[3], a[4], a[5], _p(flat), _stream()), "pack_linear_chain")

    @staticmethod
    def unpack_linear_chain(flat, mats, padded_rows, ld):
        """the matrices (fp32, the parameters' shapes) out of a flat fp16 vector in the ffmlp layout"""
        _need(flat, torch.float16, "flat")
        if flat.numel() != sum(int(r) * int(v) for r, v in zip(padded_rows, ld)) or not flat.is_contiguous():
            raise RuntimeError("unpack_linear_chain: flat holds sum(padded_rows * ld) elements")
        a = VmBackend._chain_args(mats, padded_rows, ld)
        _check(lib().s3d_unpack_linear_chain(_p(flat), a[0], a[1], a[2], a[3], a[4], a[5], _stream()), "unpack_linear_chain")

    # False (S3D_VM_BINS=torch): keys + torch.sort + searchsorted, the A/B twin of s3d_vm_backward_bins
    native_bins = os.environ.get("S3D_VM_BINS", "native") != "torch"

    @staticmethod
    def backward_bins(x, planes, resolution, n_valid=None):
        """(perm [6,N] i32, start [6,n_bounds] i32, n_bounds): the points sorted by plane tile / line chunk, as the backward
        kernels want them.  Depends on x and the resolution only — the density and the colour factors of one network share it."""
        N, dev = x.shape[0], x.device
        u3 = C.c_uint32 * 3
        rank = u3(*[int(t.shape[1]) for t in planes])
        res = u3(*[int(r) for r in resolution])
        n_bounds = int(lib().s3d_vm_backward_max_bins(res)) + 2
        if VmBackend.native_bins:
            perm = torch.empty(6, N, dtype=torch.int32, device=dev)
            start = torch.empty(6, n_bounds, dtype=torch.int32, device=dev)
            nbytes = int(lib().s3d_vm_backward_bins_workspace_size(_u(N), _u(n_bounds)))
            work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _check(lib().s3d_vm_backward_bins(_p(x), _u(N), rank, res, _p(perm), _p(start), _u(n_bounds), _p(work), C.c_size_t(nbytes),
                                              _nv(n_valid), _stream()), "vm_backward_bins")
            return perm, start, n_bounds
        if n_valid is not None:
            raise RuntimeError("vm backward bins: the torch.sort twin (A/B) has no padded-batch form")
        keys = torch.empty(6, N, dtype=torch.int32, device=dev)
        _check(lib().s3d_vm_backward_keys(_p(x), _u(N), rank, res, _p(keys), _stream()), "vm_backward_keys")
        # (A/B path: the same list through torch.sort — one sort over all six rows, row number above the key bits)
        bits = max(int(n_bounds).bit_length(), 1)
        if bits > 27:
            raise RuntimeError("vm features backward: resolution too large for 32-bit sort keys")
        rows6 = torch.arange(6, dtype=torch.int32, device=dev).unsqueeze(1)
        skeys, order = torch.sort(((rows6 << bits) | keys.clamp_(max=(1 << bits) - 1)).view(-1), stable=True)
        bounds = (rows6 << bits) | torch.arange(n_bounds, dtype=torch.int32, device=dev).unsqueeze(0)
        start = (torch.searchsorted(skeys, bounds.view(-1)).view(6, n_bounds) - rows6 * N).to(torch.int32).contiguous()
        perm = (order.view(6, N) - rows6.long() * N).to(torch.int32).contiguous()
        return perm, start, n_bounds

    @staticmethod
    def transpose_factors(planes, lines, resolution):
        """rank-fastest shadows of a factor set (s3d_vm_transpose_factors): ([H, W, R_i] x 3, [Dn, R_i] x 3) fp32, from the CURRENT
        values of the parameters — the caller takes them afresh whenever the parameters may have changed"""
        for t in list(planes) + list(lines):
            _need(t, torch.float32, "factor")
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("vm transpose: factors must be contiguous GPU tensors")
        ptr3, u3 = C.c_void_p * 3, C.c_uint32 * 3
        pt = [torch.empty(t.shape[2], t.shape[3], t.shape[1], dtype=torch.float32, device=t.device) for t in planes]
        lt = [torch.empty(t.shape[2], t.shape[1], dtype=torch.float32, device=t.device) for t in lines]
        _check(lib().s3d_vm_transpose_factors(ptr3(*[t.data_ptr() for t in planes]), ptr3(*[t.data_ptr() for t in lines]),
                                              u3(*[int(t.shape[1]) for t in planes]), u3(*[int(r) for r in resolution]),
                                              ptr3(*[t.data_ptr() for t in pt]), ptr3(*[t.data_ptr() for t in lt]), _stream()),
               "vm_transpose_factors")
        return pt, lt

    @staticmethod
    def _stage(N, rank, res, dev):
        """staging rows of the factor backward's flushes (s3d_vm_backward_stage_bytes; uint8, no initialisation)"""
        return torch.empty(int(lib().s3d_vm_backward_stage_bytes(_u(N), rank, res)), dtype=torch.uint8, device=dev)

    @staticmethod
    def features_backward(x, planes, lines, resolution, reduce, grad, bins=None, found_inf=None, n_valid=None, shadows=None):
        """gradients of features_forward w.r.t. planes / lines (lists shaped like the factors).  grad: [N] (reduce) or
        [N, sum R_i] point-major.  `bins`: a backward_bins() result for the same x / resolution."""
        _need(x, torch.float32, "x"); _need(grad, torch.float32, "grad")
        N, dev = x.shape[0], x.device
        ptr3, u3 = C.c_void_p * 3, C.c_uint32 * 3
        rank = u3(*[int(t.shape[1]) for t in planes])
        res = u3(*[int(r) for r in resolution])
        rows = sum(int(t.shape[1]) for t in planes)
        if not grad.is_contiguous() or grad.numel() != (N if reduce else N * rows):
            raise RuntimeError("vm features backward: grad must be contiguous [N] / [N, sum rank]")
        perm, start, n_bounds = bins if bins is not None else VmBackend.backward_bins(x, planes, resolution, n_valid)
        gm = torch.empty(N, rows, dtype=torch.float32, device=dev)  # (written by the plane pass for every point the line pass reads)
        # (named locals: temporaries created inside the argument list would be freed one by one and handed the SAME block)
        gs, bound_words = _zeros_like_many(list(planes) + list(lines), 4)
        g_planes, g_lines = gs[:3], gs[3:]
        line_scratch = torch.empty(sum(t.numel() for t in lines), dtype=torch.float32, device=dev)
        stage = VmBackend._stage(N, rank, res, dev)
        _check(lib().s3d_vm_features_backward(_p(x), _u(N), ptr3(*[t.data_ptr() for t in planes]),
                                              ptr3(*[t.data_ptr() for t in lines]), rank, res, C.c_int(int(bool(reduce))),
                                              _p(grad), _p(perm), _p(start), _u(n_bounds), _p(gm),
                                              ptr3(*[t.data_ptr() for t in g_planes]), ptr3(*[t.data_ptr() for t in g_lines]),
                                              _p(bound_words), _p(line_scratch), _p(stage), C.c_size_t(stage.numel()), _p(found_inf),
                                              *_shadow1(shadows), _nv(n_valid), _stream()), "vm_features_backward")
        return g_planes, g_lines

    @staticmethod
    def color_forward(x, planes, lines, resolution, basis, out, n_valid=None, shadows=None):
        """colour products with basis_mat applied in the kernel: basis fp16 [Cb, sum R_i] (the Linear's weight), out fp16 [N, Cb]"""
        _need(x, torch.float32, "x"); _need(basis, torch.float16, "basis"); _need(out, torch.float16, "out")
        for t in list(planes) + list(lines):
            _need(t, torch.float32, "factor")
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("vm features: factors must be contiguous GPU tensors")
        rows = sum(int(t.shape[1]) for t in planes)
        if not x.is_contiguous() or x.shape[-1] != 3 or not basis.is_contiguous() or basis.shape[1] != rows:
            raise RuntimeError("vm color: x must be contiguous [N,3], basis contiguous [Cb, sum rank]")
        if not out.is_contiguous() or out.numel() != x.shape[0] * basis.shape[0]:
            raise RuntimeError("vm color: out must be contiguous [N, Cb]")
        ptr3, u3 = C.c_void_p * 3, C.c_uint32 * 3
        _check(lib().s3d_vm_color_forward(_p(x), _u(x.shape[0]), ptr3(*[t.data_ptr() for t in planes]),
                                          ptr3(*[t.data_ptr() for t in lines]), u3(*[int(t.shape[1]) for t in planes]),
                                          u3(*[int(r) for r in resolution]), _p(basis), _u(basis.shape[0]), _p(out),
                                          *_shadow2(shadows), _nv(n_valid), _stream()), "vm_color_forward")

    @staticmethod
    def color_backward(x, planes, lines, resolution, basis, grad_out, bins=None, found_inf=None, n_valid=None, shadows=None):
        """gradients of color_forward w.r.t. planes / lines / basis from grad_out fp16 [N, Cb]: (g_planes, g_lines, g_basis fp32)"""
        _need(x, torch.float32, "x"); _need(basis, torch.float16, "basis"); _need(grad_out, torch.float16, "grad_out")
        N, dev = x.shape[0], x.device
        ptr3, u3 = C.c_void_p * 3, C.c_uint32 * 3
        rows = sum(int(t.shape[1]) for t in planes)
        if grad_out.shape != (N, basis.shape[0]) or not basis.is_contiguous() or basis.shape[1] != rows or basis.shape[0] > 32:
            raise RuntimeError("vm color backward: grad_out must be [N, Cb <= 32], basis contiguous [Cb, sum rank]")
        # the kernel reads a point's gradients as four 16-byte words: rows padded to 32 columns (a [:, :Cb] view of a zero-padded
        # [N, 32] buffer — FreqBackend.freq_encode_pack_backward marks what it wrote — is taken as it is)
        base = grad_out._base
        if (base is not None and getattr(base, "_s3d_zero_padded", False) and base.shape == (N, 32) and base.is_contiguous()
                and grad_out.data_ptr() == base.data_ptr() and grad_out.stride() == (32, 1)):
            grad_out = base
        else:
            grad_out = torch.nn.functional.pad(grad_out, (0, 32 - basis.shape[0])).contiguous()
        perm, start, n_bounds = bins if bins is not None else VmBackend.backward_bins(x, planes, resolution, n_valid)
        gm = torch.empty(N, rows, dtype=torch.float32, device=dev)  # (written by the plane pass for every point the line pass reads)
        gs, bound_words = _zeros_like_many(list(planes) + list(lines) + [basis], 4)
        g_planes, g_lines, g_basis = gs[:3], gs[3:6], gs[6]
        line_scratch = torch.empty(sum(t.numel() for t in lines), dtype=torch.float32, device=dev)
        rank, res = u3(*[int(t.shape[1]) for t in planes]), u3(*[int(r) for r in resolution])
        stage = VmBackend._stage(N, rank, res, dev)
        _check(lib().s3d_vm_color_backward(_p(x), _u(N), ptr3(*[t.data_ptr() for t in planes]),
                                           ptr3(*[t.data_ptr() for t in lines]), rank, res, _p(basis), _u(basis.shape[0]), _p(grad_out),
                                           _p(perm), _p(start), _u(n_bounds), _p(gm), ptr3(*[t.data_ptr() for t in g_planes]),
                                           ptr3(*[t.data_ptr() for t in g_lines]), _p(g_basis),
                                           _p(bound_words), _p(line_scratch), _p(stage), C.c_size_t(stage.numel()), _p(found_inf),
                                           *_shadow1(shadows), _nv(n_valid), _stream()), "vm_color_backward")
        return g_planes, g_lines, g_basis
