#!/bin/bash
# GPU box: PMC passes over the isolated grid backward (ray-ordered, B = 2^18)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r08_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no_fwd --iters 6 --sizes 262144 --orders ray"
i=0
for set in "${@:-SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $ROOT/tools/bench_grid.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  for k in k_bin_scatter k_bin_accumulate6; do echo "## $k [$set]"; python $ROOT/tools/pmc_kernel.py $OUT/p$i $k 2>/dev/null; done
  rm -rf $OUT/p$i
done 2>&1 | tee $OUT/pmc.txt
