"""Seal teacher/student renderers (SealNeRF/renderer.py).

The teacher is a pretrained network viewed through the proxy function: every marched sample is mapped back to source
space before the network query (`map_samples`, SealNeRF/renderer.py:291-316 / 381-399), and the cells inside the edit
bound are forced "occupied" in the bitfield so that the (empty, in the source scene) target region is sampled at all
(`hack_bitfield`, :50-66).  The student is a plain network whose bitfield is force-filled the same way."""
import torch

import raymarching


class SealTeacherMixin:
    seal_mapper = None
    density_bitfield_origin = None
    density_bitfield_hacked = False
    proxy_enabled = True

    def init_mapper(self, mapper):
        """SealNeRF/renderer.py:22-47: cells of the force-fill bounds -> bitfield byte indices.  As there, the mapper's
        `force_fill_bound` is clamped to the inference box IN PLACE: the pretraining lattice (trainer.py: init_pretraining)
        and every later init_mapper of the same mapper (the student's) see the clamped bounds."""
        self.seal_mapper = mapper
        dev = self.density_bitfield.device
        bounds = mapper.map_data["force_fill_bound"]
        if bounds.ndim == 2:
            bounds = bounds[None]  # (a view: the clamp below still lands in map_data)
        aabb = self.aabb_infer.to(bounds.device)
        bounds[:, 0, :] = torch.max(bounds[:, 0, :], aabb[:3])
        bounds[:, 1, :] = torch.min(bounds[:, 1, :], aabb[-3:])
        idx = []
        for i in range(bounds.shape[0]):
            cmin, cmax = torch.floor(((bounds[i] + self.bound) / self.bound / 2) * self.grid_size).to(dev)
            X, Y, Z = torch.meshgrid(torch.arange(cmin[0], cmax[0], device=dev), torch.arange(cmin[1], cmax[1], device=dev),
                                     torch.arange(cmin[2], cmax[2], device=dev), indexing="ij")
            coords = torch.stack([X, Y, Z], dim=-1).reshape(-1, 3)
            idx.append(raymarching.morton3D(coords.int()).long())
        self.force_fill_grid_indices = torch.cat(idx)
        self.force_fill_bitfield_indices = self.force_fill_grid_indices // 8

    @torch.no_grad()
    def hack_bitfield(self):
        if self.density_bitfield_origin is None:
            self.density_bitfield_origin = self.density_bitfield[self.force_fill_bitfield_indices]
        self.density_bitfield[self.force_fill_bitfield_indices] = 255
        self.density_bitfield_hacked = True

    @torch.no_grad()
    def restore_bitfield(self):
        self.density_bitfield[self.force_fill_bitfield_indices] = self.density_bitfield_origin
        self.density_bitfield_hacked = False

    def update_extra_state(self, decay=0.95, S=128):
        super().update_extra_state(decay, S)
        self.after_extra_state()

    # update_extra_state = the backbone's sweep + this epilogue (nerf/trainer.py replays the sweep from a graph and calls it)
    extra_state_epilogue_only = True

    def after_extra_state(self):
        if self.seal_mapper is not None:
            self.hack_bitfield()

    def _plain_sample_path(self):
        """nerf/renderer.py: may run_cuda skip the zero fills of the sample buffers?  Yes without a proxy (student), and with
        the native bbox mapper too: it takes the device-side sample count like every other per-sample kernel (rows behind
        the count are neither read nor written; the marcher keeps the pad rows up to the next 128 zero, the mapper maps them).
        The bbox tool's colour edit runs on the device with the same count (csrc/seal.hip: s3d_seal_map_color); the brush
        tool's image remap (torch ops over whole tensors) would keep the conservative path."""
        if self.seal_mapper is None or not self.proxy_enabled:
            return True
        m = self.seal_mapper
        return bool(getattr(m, "native", False)) and "image" not in m.map_data

    def _batch_dependent_colors(self):
        return self.seal_mapper is not None and self.proxy_enabled and "rgb" in self.seal_mapper.map_data

    # teacher only: proxy the samples
    def map_samples(self, xyzs, dirs):
        if self.seal_mapper is None or not self.proxy_enabled:
            return xyzs, dirs, None
        return self.seal_mapper.map_to_origin(xyzs.view(-1, 3), dirs.view(-1, 3))

    def map_colors(self, xyzs, dirs, rgbs, mask):
        """`rgbs[mapped_mask] = map_color(mapped_xyzs[mapped_mask], mapped_dirs[mapped_mask], rgbs[mapped_mask])`
        (SealNeRF/renderer.py:316, 396-399): only the samples the proxy moved are re-coloured, and the batch statistics of a
        colour edit (modify_rgb's mean brightness, seal_utils.py:753-769) are those of the moved samples alone."""
        if mask is None or self.seal_mapper is None:
            return rgbs
        md = self.seal_mapper.map_data
        if "hsv" not in md and "rgb" not in md:
            return rgbs
        return self.seal_mapper.map_color_masked(xyzs, dirs, rgbs, mask)


def _mix(net_cls, name, proxy):
    return type(name, (SealTeacherMixin, net_cls), {"proxy_enabled": proxy})


def make_teacher(net_cls, *args, **kwargs):
    """teacher = backbone network + proxy mapping (SealNeRF/network.py:7-46 builds these classes dynamically)"""
    return _mix(net_cls, "SealTeacher" + net_cls.__name__, True)(*args, **kwargs)


def make_student(net_cls, *args, **kwargs):
    return _mix(net_cls, "SealStudent" + net_cls.__name__, False)(*args, **kwargs)
