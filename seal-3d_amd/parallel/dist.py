"""Ray/point-sharded data parallelism for the distillation / training step.

The reference has no live multi-GPU path (dead DDP hooks only, nerf/utils.py:330-333).  The path shards
naturally: every rank holds a full replica (tables 49 MB fp16, MLPs, bitfield), marches and shades its own
slice of the rays (or pretraining points), and the only exchange is ONE sum all-reduce per step over a flat
fp32 gradient bucket (hash-table + MLP grads, 98 MB for the Seal NGP net) — RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests.  Parameter `.grad`s are views into the bucket, so backward writes
straight into it and no gather/scatter copies are needed.  GradScaler consistency: gradients are reduced
BEFORE `unscale_`, so an overflow on any rank makes the reduced gradient non-finite on every rank and all
replicas skip the step together.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_slice(n, rank, world):
    """contiguous shard [lo, hi) of n units for `rank` (remainder spread over the first ranks)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RayShardedDP:
    def __init__(self, group=None, average=True, force_collective=False):
        self.group = group
        self.force_collective = force_collective  # issue the collectives even for world == 1 (single-GPU test of the path)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.average = average
        self._avg_ok = None
        self.flat = None
        self.params = []
        self.half_grads = []

    def register(self, model):
        """broadcast rank 0's replica and re-home every parameter gradient inside one flat bucket"""
        # tables whose gradient is handed over as an fp16 buffer (nerf/optim.py) are reduced in that buffer — half the
        # bytes on the wire; everything else is re-homed inside the flat fp32 bucket
        self.half_grads, seen = [], set()
        for p in model.parameters():
            if p.requires_grad and getattr(p, "_s3d_grad", None) is not None:
                buf = getattr(p, "_s3d_grad_flat", p._s3d_grad)  # the optimizer's single flat buffer when there is one
                if buf.data_ptr() not in seen:
                    seen.add(buf.data_ptr())
                    self.half_grads.append(buf)
        self.params = [p for p in model.parameters() if p.requires_grad and getattr(p, "_s3d_grad", None) is None]
        if self.world > 1:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)
        total = sum(p.numel() for p in self.params)
        dev = next(model.parameters()).device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        if self.average and (self.world > 1 or self.force_collective):
            self._avg_supported()  # probe now (host read-back): never inside a graph capture
        return self

    def _avg_supported(self):
        """ReduceOp.AVG on this group, probed once with a real (tiny) fp16 collective: every rank takes part, so every rank
        reaches the same verdict; anything but the exact mean falls back to SUM + divide"""
        if self._avg_ok is None:
            ok = False
            if dist.get_backend(self.group) == "nccl":
                try:
                    probe = torch.full((8,), 3.0, dtype=torch.float16, device="cuda")
                    dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.group)
                    ok = bool((probe == 3.0).all().item())
                except RuntimeError:
                    ok = False
            self._avg_ok = ok
        return self._avg_ok

    def allreduce_grads(self, scaler=None):
        if self.world == 1 and not self.force_collective:
            return
        # safety: a grad that autograd re-allocated is copied back into its bucket slot
        off = 0
        for p in self.params:
            n = p.numel()
            slot = self.flat[off:off + n]
            if p.grad is None:
                slot.zero_()
                p.grad = slot.view_as(p)
            elif p.grad.data_ptr() != slot.data_ptr():
                slot.copy_(p.grad.reshape(-1))
                p.grad = slot.view_as(p)
            off += n
        # RCCL averages inside the collective (pre-multiplied sum): no separate divide pass over the bucket, and the fp16
        # bucket of loss-scaled gradients cannot overflow in the sum of `world` ranks.  gloo (CPU tests) has no AVG.
        fused_avg = self.average and self._avg_supported()
        op = dist.ReduceOp.AVG if fused_avg else dist.ReduceOp.SUM
        for h in self.half_grads:
            dist.all_reduce(h, op=op, group=self.group)
            if self.average and not fused_avg and self.world > 1:
                h.div_(self.world)
        if self.flat.numel():
            dist.all_reduce(self.flat, op=op, group=self.group)
            if self.average and not fused_avg:
                self.flat.div_(self.world)

    def sync_extra_state(self, model):
        """keep the occupancy state identical on all replicas after `update_extra_state` (RNG differs per rank)"""
        if (self.world == 1 and not self.force_collective) or not getattr(model, "cuda_ray", False):
            return
        dist.broadcast(model.density_grid, src=0, group=self.group)
        dist.broadcast(model.density_bitfield, src=0, group=self.group)
        # sample budget: the largest running mean over the ranks (each rank marches its own rays), density: rank 0's
        t = torch.tensor([float(model.mean_count), float(model.mean_density) if self.rank == 0 else float("-inf")],
                         device=model.density_grid.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        model.mean_count, model.mean_density = int(t[0].item()), float(t[1].item())

    def all_reduce_scalar(self, value, op="sum"):
        if self.world == 1:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.flat.device if self.flat is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=self.group)
        return float(t.item())
