#!/bin/bash
# GPU box: per-kernel durations of the isolated grid backward (ray-ordered, B = 2^18) for the product library and for every
# variant named on the command line (tools/build_variants.sh): tools/trace_bwd.sh [variant ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in product "$@"; do
  if [ "$v" = product ]; then unset S3D_HIP_LIB; else export S3D_HIP_LIB=$ROOT/seal-3d_amd/csrc/build/variants/lib_$v.so; fi
  rm -rf /tmp/trace_bwd
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_bwd -- python $ROOT/tools/bench_grid.py --no_fwd --iters 30 --sizes 262144 --orders ${ORDERS:-ray} > /tmp/trace_bwd.log 2>&1
  echo "== $v: $(grep grid_bwd /tmp/trace_bwd.log | tr -s ' ' | cut -c1-90)"
  python $ROOT/tools/kstats.py /tmp/trace_bwd 5.0 2>&1 | grep k_bin
done
