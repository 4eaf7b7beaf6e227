#!/bin/bash
# GPU box: PMC passes (own run each, --kernel-trace only beside --pmc) over the TensoRF training step at resolution 300 for the
# factor-backward kernels -> gpurun_out/${S3D_PMC_TAG:-r10}_pmc_tensorf.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r10_pmc_tf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $ROOT/tools/bench_tensorf_step.py 300 fused native > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  for k in ${S3D_PMC_KERNELS:-"k_vm_plane_backward<64" "k_vm_plane_backward<16" "k_vm_line_backward<64"}; do echo "## $k [$set]"; python $ROOT/tools/pmc_kernel.py $OUT/p$i $k 2>/dev/null; done
  rm -rf $OUT/p$i
done 2>&1 | tee $ROOT/gpurun_out/${S3D_PMC_TAG:-r10}_pmc_tensorf.txt
