#!/bin/bash
# GPU box: kernel averages of the TensoRF factor backward (32-points-per-trip kernels) for several S3D_VM_MM_PTS settings
# (plane48,plane16,line48,line16) of ONE library (S3D_HIP_LIB or the in-tree one).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  rm -rf /tmp/vm_sweep
  S3D_VM_MM_PTS=$t timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vm_sweep -- python "$ROOT/tools/bench_tensorf_step.py" 300 fused native > /tmp/vm_sweep.log 2>&1
  F=$(find /tmp/vm_sweep -name "*kernel_stats.csv" | head -1)
  echo "S3D_VM_MM_PTS=$t  $(grep 'ms/step' /tmp/vm_sweep.log | tail -1 | sed 's/.*trainer: *//')"
  python - "$F" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_vm_" in r["Name"] and "backward" in r["Name"]:
        m = re.search(r"(k_vm_\w+<[^>]*>)", r["Name"])
        print(f"    {(m.group(1) if m else r['Name'][-60:]):44s} {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']}")
PY
done
