#!/bin/bash
# Runs on the GPU box (via gpurun): BASELINE configs[4] (TensoRF VM-48) training step, fused VM kernels vs the reference's
# grid_sample sequence, plus a kernel trace of the fused step at resolution 300 -> gpurun_out/profiles_out/<tag>_tensorf.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r09}
mkdir -p "$ROOT/gpurun_out/profiles_out"
cd /tmp && export TMPDIR=/tmp
# $2 = "quick": only the fused kernels at resolution 300 (the grid_sample sequence takes 15 - 17 ms per step)
if [ "${2:-full}" = "quick" ]; then
  python "$ROOT/tools/bench_tensorf_step.py" 300 fused torch,native,graph > /tmp/tensorf_steps.log 2>&1 || tail -5 /tmp/tensorf_steps.log
else
  python "$ROOT/tools/bench_tensorf_step.py" 128,300 fused,torch torch,native,graph > /tmp/tensorf_steps.log 2>&1 || tail -5 /tmp/tensorf_steps.log
fi
rm -rf /tmp/tensorf_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tensorf_prof -- python "$ROOT/tools/bench_tensorf_step.py" 300 fused graph > /tmp/tensorf_prof.log 2>&1
F=$(find /tmp/tensorf_prof -name "*kernel_stats.csv" | head -1)
python - "$F" "$TAG" > "$ROOT/gpurun_out/profiles_out/${TAG}_tensorf.md" <<'PY'
import csv, sys
f, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 kernel trace `{tag}` — TensoRF VM-48 training step (BASELINE configs[4]; `tools/bench_tensorf_step.py`, `tools/profile_tensorf.sh`)\n")
print("4,096 rays per step on the synthetic Lego-shaped scene, trainers of tensoRF/utils.py (torch = eager, torch.optim.Adam + GradScaler; native = eager, NativeAdam; graph = HIP-graph replay, NativeAdam), every step with the L1 penalty of the reference's step, marching / compositing by the build's kernels.\n")
print("```")
print(open("/tmp/tensorf_steps.log").read().strip())
print("```\n")
print(f"Kernel trace of the whole process at resolution 300, fused VM kernels, graph trainer (warm-up + timed steps: 24 executions of every kernel of the step; {tot/1e6:.1f} ms of kernels).  Share of kernel time:\n")
print("| kernel | calls | us/call | % of kernel time |")
print("|---|---|---|---|")
for r in rows[:32]:
    n = r["Name"].replace("void ", "").replace("s3d::(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:90]
    print(f"| `{n}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
PY
cat "$ROOT/gpurun_out/profiles_out/${TAG}_tensorf.md"
