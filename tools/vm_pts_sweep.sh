#!/bin/bash
# Runs on the GPU box: kernel averages of the TensoRF factor backward for several S3D_VM_PTS settings, same box, same process
# sequence (box-to-box differences are larger than the differences between the settings).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for p in "$@"; do
  rm -rf /tmp/vm_sweep
  S3D_VM_PTS=$p timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vm_sweep -- python "$ROOT/tools/bench_tensorf_step.py" 300 fused native > /tmp/vm_sweep.log 2>&1
  F=$(find /tmp/vm_sweep -name "*kernel_stats.csv" | head -1)
  echo "S3D_VM_PTS=$p  $(grep 'ms/step' /tmp/vm_sweep.log | tail -1 | sed 's/.*trainer: *//')"
  python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_vm_" in r["Name"]:
        n = r["Name"].split("(")[0].replace("void s3d::(anonymous namespace)::", "")[:48]
        print(f"    {n:50s} {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']}")
PY
done
