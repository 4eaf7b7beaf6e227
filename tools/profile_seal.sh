#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the Seal distillation phases -> gpurun_out/profiles_out/<tag>_seal.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r08}
mkdir -p "$ROOT/gpurun_out/profiles_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/seal_prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/seal_prof -- python "$ROOT/tools/seal_phases.py" > /tmp/seal_prof.log 2>&1 || tail -5 /tmp/seal_prof.log
python "$ROOT/tools/seal_phases.py" --summarize /tmp/seal_prof "$TAG" > "$ROOT/gpurun_out/profiles_out/${TAG}_seal.md"
head -60 "$ROOT/gpurun_out/profiles_out/${TAG}_seal.md"
