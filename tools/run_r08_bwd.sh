#!/bin/bash
# GPU box: correctness of the third-generation binned backward + A/B of its variants (tools/build_variants.sh)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r08_bwd; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_padded_batch.py tests/test_gpu_golden.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
echo "== gen2 (path 3)"; timeout 300 python tools/bench_grid.py --no_fwd --sum --iters 60 --sizes 262144 --bwd_path 3 2>&1 | grep grid_bwd
echo "== gen3 default"; timeout 300 python tools/bench_grid.py --no_fwd --sum --iters 60 --sizes 262144 2097152 2>&1 | tee $OUT/gen3.log | grep grid_bwd
for v in "$@"; do
  echo "== $v"; S3D_HIP_LIB=$ROOT/seal-3d_amd/csrc/build/variants/lib_$v.so timeout 300 python tools/bench_grid.py --no_fwd --sum --iters 60 --sizes 262144 2>&1 | tee $OUT/$v.log | grep grid_bwd
done
if [ -f $ROOT/seal-3d_amd/csrc/build/variants/lib_prof.so ]; then
  S3D_HIP_LIB=$ROOT/seal-3d_amd/csrc/build/variants/lib_prof.so timeout 300 python tools/prof_bwd_phases.py ray 2>&1 | tee $OUT/prof_ray.log
fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/tools/bench_grid.py --no_fwd --iters 10 --sizes 262144 --orders ray > $OUT/trace.log 2>&1
python $ROOT/tools/kstats.py $OUT/trace 1.0 2>&1 | head -8 | tee $OUT/stats.txt
rm -rf $OUT/trace
