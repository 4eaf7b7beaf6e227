#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE passes over the TensoRF step -> gpurun_out/<tag>_tensorf_pmc.json (tools/pmc_tensorf_traffic.py)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r11}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tf_fetch /tmp/tf_write
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tf_fetch -- python $ROOT/tools/bench_tensorf_step.py 300 fused native > /tmp/tf_fetch.log 2>&1 || tail -3 /tmp/tf_fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tf_write -- python $ROOT/tools/bench_tensorf_step.py 300 fused native > /tmp/tf_write.log 2>&1 || tail -3 /tmp/tf_write.log
python $ROOT/tools/pmc_tensorf_traffic.py /tmp/tf_fetch /tmp/tf_write > $ROOT/gpurun_out/${TAG}_tensorf_pmc.json
cat $ROOT/gpurun_out/${TAG}_tensorf_pmc.json
