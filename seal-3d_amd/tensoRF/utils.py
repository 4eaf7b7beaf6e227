"""TensoRF trainer (tensoRF/utils.py:14-153): the NGP training step plus the L1 penalty on the density factors, two learning
rates (factors `lr0`, networks `lr1`: main_SealTensoRF.py:30-33, tensoRF/network.py:322-331) and the resolution schedule —
at every step listed in `upsample_model_steps` the model is cropped to the occupied box (`shrink_model`), its factors are
re-sampled to the next resolution (voxel count `upsample_resolutions[k] ** 3` spread over the cropped box) and the optimizer
is re-created over the new parameters (tensoRF/utils.py:124-140)."""
import numpy as np
import torch

from nerf.trainer import GraphedTrainer as _GraphedTrainer
from nerf.trainer import Trainer as _Trainer


class TensoRFSteps:
    """mixed into nerf.trainer.Trainer / GraphedTrainer"""

    def _init_tensorf(self, lr0, lr1, l1_reg_weight, upsample_model_steps, upsample_resolutions):
        self.lr0, self.lr1 = lr0, lr1
        self.l1_reg_weight = l1_reg_weight
        self.upsample_model_steps = list(upsample_model_steps)
        self.upsample_resolutions = list(upsample_resolutions)

    def _param_groups(self):
        return self.model.get_params(self.lr0, self.lr1)

    def _attach_source_checks(self):
        """single replica + native GradScaler: the factor backward raises the scaler's flag itself (s3d_vm_*_backward(found_inf))
        and the scaler's check pass skips the 69 MB of factor gradients; with a data-parallel layer the REDUCED gradient is what
        has to be checked, so nothing is attached"""
        flag = getattr(self.scaler, "_found_inf", None)
        ok = flag is not None and getattr(self.scaler, "enabled", False) and self.dist is None and self.native_optim
        self.model.__dict__["_s3d_found_inf"] = flag if ok else None

    l1_in_update = True  # native optimizer: the penalty's gradient is formed inside the Adam launch (False: autograd, A/B + tests)

    def _regularizer(self):
        """`density_loss() * l1_reg_weight` (tensoRF/utils.py:42-49).  With the native optimizer the term enters the loss as a
        VALUE and its gradient — l1_reg_weight / numel * sign(factor) — is added to the unscaled gradient inside NativeAdam's
        launch (s3d_adam_tensor.l1; announced per step on the factors, consumed by the step): the sign, two scalings and six
        accumulate passes over the density factors of the autograd route are ~0.1 ms of a 2.1 ms step."""
        if not self.l1_reg_weight:
            return None
        from nerf.optim import NativeAdam
        if (self.l1_in_update and self.native_optim and self.dist is None and isinstance(self.optimizer, NativeAdam)
                and self._expected_grad() is not None):
            for p in list(self.model.sigma_mat) + list(self.model.sigma_vec):
                p._s3d_l1 = self.l1_reg_weight / p.numel() if p.requires_grad else 0.0
            return self.model.density_loss_value() * self.l1_reg_weight
        return self.model.density_loss() * self.l1_reg_weight

    def next_resolution(self):
        """adaptive voxel size from the (cropped) training box (tensoRF/utils.py:128-133)"""
        n_vox = self.upsample_resolutions.pop(0) ** 3
        aabb = self.model.aabb_train.detach().cpu().numpy()
        vox = np.cbrt(np.prod(aabb[3:] - aabb[:3]) / n_vox)
        return ((aabb[3:] - aabb[:3]) / vox).astype(np.int32).tolist()

    def _maybe_upsample(self):
        if self.global_step not in self.upsample_model_steps:
            return False
        if self.model.cuda_ray:
            self.model.shrink_model()
        self.model.upsample_model(self.next_resolution())
        self.rebuild_optimizer()  # (the parameter set changed; the reference re-creates optimizer and scheduler too)
        if hasattr(self, "graph"):
            self.graph = self.graph_opt = None  # a captured step holds the old factors' pointers
        return True


class Trainer(TensoRFSteps, _Trainer):
    def __init__(self, model, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, upsample_model_steps=(), upsample_resolutions=(), **kw):
        self._init_tensorf(lr0, lr1, l1_reg_weight, upsample_model_steps, upsample_resolutions)
        _Trainer.__init__(self, model, lr=lr0, **kw)
        self._attach_source_checks()

    def train_step(self, rays_o, rays_d, gt_rgb, bg_color=1):
        loss = _Trainer.train_step(self, rays_o, rays_d, gt_rgb, bg_color)
        self._maybe_upsample()
        return loss


class GraphedTrainer(TensoRFSteps, _GraphedTrainer):
    """the same step replayed from a HIP graph (nerf/trainer.py: GraphedTrainer); re-captured after every upsampling"""

    def __init__(self, model, num_rays, lr0=2e-2, lr1=1e-3, l1_reg_weight=1e-4, upsample_model_steps=(), upsample_resolutions=(), **kw):
        self._init_tensorf(lr0, lr1, l1_reg_weight, upsample_model_steps, upsample_resolutions)
        # the static sample budget: since round 6 every kernel of this backbone's sample path takes the device-side row count
        # (s3d_hip.row_limit -> n_valid), so the padding costs nothing and the budget keeps the NGP trainer's 30 % of headroom over
        # the running mean (the reference's own budget is the running mean itself, nerf/renderer.py:352-356; rays that do not
        # fit are dropped the same way, raymarching.cu:416)
        _GraphedTrainer.__init__(self, model, num_rays, lr=lr0, **kw)
        self._attach_source_checks()

    def train_step(self, rays_o, rays_d, gt_rgb, bg_color=1):
        loss = _GraphedTrainer.train_step(self, rays_o, rays_d, gt_rgb, bg_color)
        self._maybe_upsample()
        return loss
