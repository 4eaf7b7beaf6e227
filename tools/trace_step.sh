#!/bin/bash
# GPU box: per-kernel averages of the default bench step for several library variants on ONE box.
#   tools/trace_step.sh "<grep pattern>" product base queue base queue     (variants built by tools/build_variants.sh)
ROOT=$GRAFT_REPO_ROOT
PAT=${1:-"march\|grid_forward\|scatter"}; shift
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  if [ $v = product ]; then unset S3D_HIP_LIB; else export S3D_HIP_LIB=$ROOT/seal-3d_amd/csrc/build/variants/lib_$v.so; fi
  rm -rf /tmp/tr_$i
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$i -- python $ROOT/bench.py --steps 32 --warmup 8 --no_cpu_baseline --no_render --no_seal --no_long_run --no_tensorf > /tmp/tr_$i.log 2>&1
  echo "== $v  $(tail -1 /tmp/tr_$i.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])' 2>/dev/null)"
  python $ROOT/tools/kstats.py /tmp/tr_$i 0.5 | grep -i "$PAT"
done
