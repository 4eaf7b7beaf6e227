"""800x800 inference render (nerf/renderer.py:323-372 of the reference: march_rays / network / composite_rays loop) of a
briefly trained NGP model on the synthetic scene: wall time per frame, loop iterations, samples.  Run under
`rocprofv3 --kernel-trace --stats` for the kernel split."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import raymarching.raymarching as rm  # noqa: E402
from nerf import synthetic as syn  # noqa: E402
from nerf.trainer import GraphedTrainer  # noqa: E402
from test_gpu_trainer import _setup, _run  # noqa: E402


def main():
    scales = [int(v) for v in sys.argv[1:]] or [4]
    model, batches = _setup(n_rays=4096, n_batches=16)
    tr = GraphedTrainer(model, 4096, lr=1e-2, fp16=True)
    _run(tr, batches, 400)
    for scale in scales:
        for every in (1, 4, 8):
            model.sync_every = every
            _one(model, tr, scale)


def _one(model, tr, scale):
    model.infer_batch_scale = scale
    poses = syn.orbit_poses(1, seed=0).cuda()
    r = syn.get_rays(poses[:1], syn.lego_intrinsics(), 800, 800)
    ro, rd = r["rays_o"].contiguous(), r["rays_d"].contiguous()
    calls = {"n": 0, "pts": 0}
    inner = rm.march_rays

    def counted(n_alive, n_step, *a, **k):
        calls["n"] += 1
        calls["pts"] += n_alive * n_step
        return inner(n_alive, n_step, *a, **k)
    rm.march_rays = counted
    import nerf.renderer as nr
    nr.raymarching.march_rays = counted
    tr.render_image(ro, rd)
    torch.cuda.synchronize()
    calls["n"] = calls["pts"] = 0
    t0 = time.perf_counter()
    nfr = 5
    for _ in range(nfr):
        tr.render_image(ro, rd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nfr
    rm.march_rays = inner
    nr.raymarching.march_rays = inner
    print(f"render 800x800 infer_batch_scale={scale} sync_every={model.sync_every}: {dt*1e3:.2f} ms/frame, {0.64/dt:.1f} Mrays/s, "
          f"{calls['n']/nfr:.0f} loop iterations, {calls['pts']/nfr/1e6:.2f} M sample slots per frame", flush=True)


if __name__ == "__main__":
    main()
