#!/bin/bash
# persistent (level, chunk) queue of the hash-grid forward: S3D_FWD_SLOTS chunk slots per XCD vs one workgroup per chunk
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
{
for slots in 0 128 256 512 1024 2048; do
  echo "== S3D_FWD_SLOTS=$slots"
  S3D_FWD_SLOTS=$slots python tools/bench_grid.py --iters 30 --no_bwd --sum 2>&1 | grep grid_fwd
done
echo "== backward (plain pair), default library"
python tools/bench_grid.py --iters 20 --no_fwd 2>&1 | grep grid_bwd
echo "== cold (L2 / MALL thrashed before every launch), default"
python tools/bench_grid.py --iters 12 --cold --sizes 262144 2>&1 | grep grid_
echo "== backward + Adam"
python tools/bench_grid_adam.py; S3D_RAYS=10000 python tools/bench_grid_adam.py
} > gpurun_out/fwd_queue_sweep.log 2>&1
cat gpurun_out/fwd_queue_sweep.log
