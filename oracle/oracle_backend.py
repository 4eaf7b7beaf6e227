"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  It exposes five objects whose methods carry the
names and argument order of the reference's pybind ``_backend`` modules
(raymarching/src/bindings.cpp:6-17, gridencoder/src/bindings.cpp:6-8,
shencoder/src/bindings.cpp:6-7, freqencoder/src/bindings.cpp:6-7,
ffmlp/src/bindings.cpp:6-10) and operate on CPU torch tensors, so the same
Python wrappers that drive the HIP library can be driven by the oracle inside
a test, and — in the authoring container only — the reference's own Python
wrappers can be imported on top of it to generate golden vectors.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libs3d_oracle.so")

F32, F16 = 0, 1


def build(force=False):
    """Compile oracle/src/*.c with gcc (no-op when the .so is fresh)."""
    srcs = [os.path.join(_HERE, "src", f) for f in os.listdir(os.path.join(_HERE, "src"))]
    if not force and os.path.exists(_LIB_PATH):
        if os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.s3o_get_max_threads.restype = C.c_int
        _lib.s3o_grid_index.restype = C.c_uint32
    return _lib


def set_threads(n):
    lib().s3o_set_threads(C.c_int(int(n)))


def max_threads():
    return int(lib().s3o_get_max_threads())


def _p(t):
    """void* of a CPU tensor / numpy array / None."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    assert t.device.type == "cpu", "oracle works on CPU tensors"
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _u(x):
    return C.c_uint32(int(x))


def _f(x):
    return C.c_float(float(x))


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise RuntimeError(f"unsupported dtype {t.dtype}")


def level_scales(L, S, H):
    """Host-side per-level scale table (float32), shared by oracle and HIP."""
    out = np.empty(int(L), dtype=np.float32)
    lib().s3o_grid_level_scales(_u(L), _f(S), _u(H), _p(out))
    return out


def grid_index(D, Cc, gridtype, align_corners, ch, hashmap_size, resolution, pos_grid):
    pg = np.ascontiguousarray(pos_grid, dtype=np.uint32)
    return int(lib().s3o_grid_index(_u(D), _u(Cc), _u(gridtype), C.c_int(int(align_corners)), _u(ch),
                                    _u(hashmap_size), _u(resolution), _p(pg)))


def mip_from_pos(xyz, max_cascade):
    """raymarching.cu:42-47 on an [N,3] float32 array (test hook)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    out = np.empty(xyz.shape[0], dtype=np.int32)
    lib().s3o_mip_from_pos(_p(xyz), _u(xyz.shape[0]), _f(max_cascade), _p(out))
    return out


def mip_from_dt(dt, H, max_cascade):
    """raymarching.cu:49-54 on an [N] float32 array (test hook)."""
    dt = np.ascontiguousarray(dt, dtype=np.float32)
    out = np.empty(dt.shape[0], dtype=np.int32)
    lib().s3o_mip_from_dt(_p(dt), _u(dt.shape[0]), _f(H), _f(max_cascade), _p(out))
    return out


class RaymarchingBackend:
    """raymarching/src/raymarching.h:7-18"""
    device_type = "cpu"

    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        lib().s3o_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), _u(N), _f(min_near), _p(nears), _p(fars))

    @staticmethod
    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        lib().s3o_sph_from_ray(_p(rays_o), _p(rays_d), _f(radius), _u(N), _p(coords))

    @staticmethod
    def morton3D(coords, N, indices):
        lib().s3o_morton3D(_p(coords), _u(N), _p(indices))

    @staticmethod
    def morton3D_invert(indices, N, coords):
        lib().s3o_morton3D_invert(_p(indices), _u(N), _p(coords))

    @staticmethod
    def packbits(grid, N, density_thresh, bitfield):
        lib().s3o_packbits(_p(grid), _u(N), _f(density_thresh), _p(bitfield))

    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, H, M, nears, fars,
                         xyzs, dirs, deltas, rays, counter, noises):
        lib().s3o_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), _f(bound), _f(dt_gamma), _u(max_steps),
                                   _u(N), _u(Cc), _u(H), _u(M), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                                   _p(deltas), _p(rays), _p(counter), _p(noises))

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        lib().s3o_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), _u(M), _u(N),
                                               _f(T_thresh), _p(weights_sum), _p(depth), _p(image))

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum,
                                      image, M, N, T_thresh, grad_sigmas, grad_rgbs):
        lib().s3o_composite_rays_train_backward(_p(grad_weights_sum), _p(grad_image), _p(sigmas), _p(rgbs),
                                                _p(deltas), _p(rays), _p(weights_sum), _p(image), _u(M), _u(N),
                                                _f(T_thresh), _p(grad_sigmas), _p(grad_rgbs))

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H,
                   grid, nears, fars, xyzs, dirs, deltas, noises):
        lib().s3o_march_rays(_u(n_alive), _u(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d),
                             _f(bound), _f(dt_gamma), _u(max_steps), _u(Cc), _u(H), _p(grid), _p(nears),
                             _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(noises))

    @staticmethod
    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum,
                       depth, image):
        lib().s3o_composite_rays(_u(n_alive), _u(n_step), _f(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas),
                                 _p(rgbs), _p(deltas), _p(weights_sum), _p(depth), _p(image))


class GridBackend:
    """gridencoder/src/gridencoder.h:12-15"""
    device_type = "cpu"

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, dy_dx, gridtype,
                            align_corners, interp, corner_idx=None):
        assert inputs.dtype == torch.float32
        scales = level_scales(L, S, H)
        lib().s3o_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), _u(B), _u(D),
                                      _u(Cc), _u(L), _p(scales), _p(dy_dx), _u(gridtype),
                                      C.c_int(int(align_corners)), _u(interp), C.c_int(_dt(embeddings)),
                                      _p(corner_idx))

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, dy_dx,
                             grad_inputs, gridtype, align_corners, interp):
        scales = level_scales(L, S, H)
        lib().s3o_grid_encode_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings),
                                       _u(B), _u(D), _u(Cc), _u(L), _p(scales), _p(dy_dx), _p(grad_inputs),
                                       _u(gridtype), C.c_int(int(align_corners)), _u(interp),
                                       C.c_int(_dt(grad)))

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, Cc, L, S, H, gridtype,
                             align_corners):
        assert embeddings.dtype == torch.float32
        scales = level_scales(L, S, H)
        lib().s3o_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), _f(weight), _u(B),
                                       _u(D), _u(Cc), _u(L), _p(scales), _u(gridtype),
                                       C.c_int(int(align_corners)))


class SHBackend:
    """shencoder/src/shencoder.h:9-10"""
    device_type = "cpu"

    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, Cc, dy_dx):
        lib().s3o_sh_encode_forward(_p(inputs), _p(outputs), _u(B), _u(D), _u(Cc), _p(dy_dx))

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, Cc, dy_dx, grad_inputs):
        lib().s3o_sh_encode_backward(_p(grad), _p(inputs), _u(B), _u(D), _u(Cc), _p(dy_dx), _p(grad_inputs))


class FreqBackend:
    """freqencoder/src/freqencoder.h:7,10"""
    device_type = "cpu"

    @staticmethod
    def freq_encode_forward(inputs, B, D, deg, Cc, outputs):
        lib().s3o_freq_encode_forward(_p(inputs), _u(B), _u(D), _u(deg), _u(Cc), _p(outputs))

    @staticmethod
    def freq_encode_backward(grad, outputs, B, D, deg, Cc, grad_inputs):
        lib().s3o_freq_encode_backward(_p(grad), _p(outputs), _u(B), _u(D), _u(deg), _u(Cc), _p(grad_inputs))


class FFMLPBackend:
    """ffmlp/src/ffmlp.h:8-14"""
    device_type = "cpu"

    @staticmethod
    def allocate_splitk(n):
        pass

    @staticmethod
    def free_splitk():
        pass

    @staticmethod
    def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                      output_activation, forward_buffer, outputs):
        lib().s3o_ffmlp_forward(_p(inputs), _p(weights), _u(B), _u(input_dim), _u(output_dim), _u(hidden_dim),
                                _u(num_layers), _u(activation), _u(output_activation), _p(forward_buffer),
                                _p(outputs))

    @staticmethod
    def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                        output_activation, inference_buffer, outputs):
        lib().s3o_ffmlp_forward(_p(inputs), _p(weights), _u(B), _u(input_dim), _u(output_dim), _u(hidden_dim),
                                _u(num_layers), _u(activation), _u(output_activation), _p(None), _p(outputs))

    @staticmethod
    def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers,
                       activation, output_activation, calc_grad_inputs, backward_buffer, grad_inputs,
                       grad_weights):
        gw32 = torch.zeros(grad_weights.numel(), dtype=torch.float32)
        lib().s3o_ffmlp_backward(_p(grad), _p(inputs), _p(weights), _p(forward_buffer), _u(B), _u(input_dim),
                                 _u(output_dim), _u(hidden_dim), _u(num_layers), _u(activation),
                                 _p(backward_buffer), _p(grad_inputs if calc_grad_inputs else None), _p(gw32))
        grad_weights.copy_(gw32.to(grad_weights.dtype))
        return gw32
