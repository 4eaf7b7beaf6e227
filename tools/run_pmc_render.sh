#!/bin/bash
# GPU box: PMC passes over the render-only workload (tools/render_frames.py), per kernel of the inference loop
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/render_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python "$ROOT/tools/render_frames.py" --save /tmp/s3d_model.pth > /tmp/render_train.log 2>&1 || { tail -5 /tmp/render_train.log; exit 1; }
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $ROOT/tools/render_frames.py --load /tmp/s3d_model.pth --frames 2 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  for k in ${KERNELS:-k_march_rays k_composite_rays k_ffmlp_forward k_grid_forward_pair}; do echo "## $k [$set]"; python $ROOT/tools/pmc_kernel.py $OUT/p$i $k 2>/dev/null; done
  rm -rf $OUT/p$i
done 2>&1 | tee $OUT/pmc.txt
