#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in product nosc orig; do
  if [ $v = product ]; then unset S3D_HIP_LIB; else export S3D_HIP_LIB=$ROOT/seal-3d_amd/csrc/build/variants/lib_$v.so; fi
  rm -rf /tmp/tr_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python $ROOT/bench.py --steps 32 --warmup 8 --no_cpu_baseline --no_render --no_seal --no_long_run > /tmp/tr_$v.log 2>&1
  echo "== $v"; python $ROOT/tools/kstats.py /tmp/tr_$v 2.0 | grep -i "march\|grid_forward\|scatter"
done
