#!/bin/bash
# GPU box: kernel averages of the TensoRF factor backward for several library variants (S3D_HIP_LIB), same box.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  rm -rf /tmp/vm_sweep
  S3D_HIP_LIB=$ROOT/seal-3d_amd/csrc/build/variants/lib_$t.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vm_sweep -- python "$ROOT/tools/bench_tensorf_step.py" 300 fused native > /tmp/vm_sweep.log 2>&1
  F=$(find /tmp/vm_sweep -name "*kernel_stats.csv" | head -1)
  echo "variant $t  $(grep 'ms/step' /tmp/vm_sweep.log | tail -1 | sed 's/.*trainer: *//')"
  python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_vm_" in r["Name"] and "backward" in r["Name"]:
        n = r["Name"].split("(")[-2].split("::")[-1] if "(" in r["Name"] else r["Name"]
        print(f"    {r['Name'][-70:]:72s} {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']}")
PY
done
