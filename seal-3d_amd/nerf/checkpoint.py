"""Checkpoints in the reference's on-disk format (nerf/utils.py:1015-1137 of the reference): a plain `torch.save` dict

    {epoch, global_step, stats, [mean_count, mean_density], model: state_dict, [optimizer, lr_scheduler, scaler, ema]}

written to `{workspace}/checkpoints/{name}_ep{epoch:04d}.pth`.  The model's state-dict keys are the reference's
(`encoder.embeddings`, `encoder.offsets`, `sigma_net.weights` / `sigma_net.{i}.weight`, `color_net.*`, `density_grid`,
`density_bitfield`, `step_counter`, `aabb_train`, `aabb_infer`), so a file written by the reference's Trainer loads
here and the other way round; the Seal student starts from the teacher's file the same way (`--ckpt`).

`optimizer` holds torch.optim.Adam's layout (per-parameter `step`, `exp_avg`, `exp_avg_sq`) and `scaler`
torch.amp.GradScaler's (`scale`, `growth_factor`, `backoff_factor`, `growth_interval`, `_growth_tracker`) whichever
optimizer/scaler implementation the trainer runs (nerf/optim.py translates).
"""
import glob
import os

import torch


def checkpoint_path(workspace, name, epoch):
    return os.path.join(workspace, "checkpoints", f"{name}_ep{epoch:04d}.pth")


def save_checkpoint(trainer, workspace, name="ngp", full=False, best=False, remove_old=True, max_keep_ckpt=2,
                    lr_scheduler=None, ema=None):
    """Write the trainer's state; returns the file path (None when `best` has nothing to compare)."""
    model = trainer.model
    stats = trainer.stats
    state = {"epoch": trainer.epoch, "global_step": trainer.global_step, "stats": stats}
    if model.cuda_ray:
        state["mean_count"] = model.mean_count
        state["mean_density"] = model.mean_density
    if hasattr(model, "upsample_model") and hasattr(model, "resolution"):
        state["resolution"] = list(model.resolution)  # TensoRF: factor resolution at save time (tensoRF/utils.py:236)
    if full:
        state["optimizer"] = trainer.optimizer.state_dict()
        if lr_scheduler is not None:
            state["lr_scheduler"] = lr_scheduler.state_dict()
        state["scaler"] = trainer.scaler.state_dict()
        if ema is not None:
            state["ema"] = ema.state_dict()
    ckpt_dir = os.path.join(workspace, "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    if not best:
        state["model"] = model.state_dict()
        path = checkpoint_path(workspace, name, trainer.epoch)
        if remove_old:
            stats["checkpoints"].append(path)
            if len(stats["checkpoints"]) > max_keep_ckpt:
                old = stats["checkpoints"].pop(0)
                if os.path.exists(old):
                    os.remove(old)
        torch.save(state, path)
        return path
    # "best": only when the last evaluation improved; drops density_grid (not needed to render, :1067-1068)
    if not stats["results"]:
        return None
    if stats["best_result"] is not None and stats["results"][-1] >= stats["best_result"]:
        return None
    stats["best_result"] = stats["results"][-1]
    if ema is not None:
        ema.store()
        ema.copy_to()
    state["model"] = dict(model.state_dict())
    state["model"].pop("density_grid", None)
    if ema is not None:
        ema.restore()
    path = os.path.join(ckpt_dir, f"{name}.pth")
    torch.save(state, path)
    return path


def latest_checkpoint(workspace, name="ngp"):
    files = sorted(glob.glob(os.path.join(workspace, "checkpoints", f"{name}_ep*.pth")))
    return files[-1] if files else None


def load_checkpoint(trainer, checkpoint, model_only=False, lr_scheduler=None, ema=None):
    """Load a reference-format file into the trainer; returns (missing_keys, unexpected_keys).

    A bare state-dict (no 'model' key) is accepted like the reference does (:1081-1084)."""
    model = trainer.model
    device = next(model.parameters()).device
    ckpt = torch.load(checkpoint, map_location=device, weights_only=False)
    if "model" not in ckpt:
        model.load_state_dict(ckpt)
        _after_weights_changed(trainer)
        return [], []
    if "resolution" in ckpt and hasattr(model, "upsample_model") and list(ckpt["resolution"]) != list(model.resolution):
        # TensoRF: bring the factors to the checkpoint's resolution before loading them, then re-create the optimizer
        # over the new Parameters (tensoRF/utils.py:347-352)
        model.upsample_model(ckpt["resolution"])
        trainer.rebuild_optimizer()
    missing, unexpected = model.load_state_dict(ckpt["model"], strict=False)
    _after_weights_changed(trainer)
    if ema is not None and "ema" in ckpt:
        ema.load_state_dict(ckpt["ema"])
    if model.cuda_ray:
        if "mean_count" in ckpt:
            model.mean_count = ckpt["mean_count"]
        if "mean_density" in ckpt:
            model.mean_density = ckpt["mean_density"]
    if model_only:
        return list(missing), list(unexpected)
    trainer.stats = ckpt["stats"]
    trainer.epoch = ckpt["epoch"]
    trainer.global_step = ckpt["global_step"]
    if trainer.optimizer is not None and "optimizer" in ckpt:
        trainer.optimizer.load_state_dict(ckpt["optimizer"])
    if lr_scheduler is not None and "lr_scheduler" in ckpt:
        lr_scheduler.load_state_dict(ckpt["lr_scheduler"])
    if trainer.scaler is not None and "scaler" in ckpt and ckpt["scaler"]:
        trainer.scaler.load_state_dict(ckpt["scaler"])
    return list(missing), list(unexpected)


def _after_weights_changed(trainer):
    """fp16 copies of the tables / MLP weights (eval cache, optimizer hand-over) follow the new fp32 values"""
    from gridencoder.grid import bump_weights_epoch
    bump_weights_epoch()
    resync = getattr(trainer.optimizer, "resync_half", None)
    if resync is not None:
        resync()
