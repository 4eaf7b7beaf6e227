#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the render-only workload -> gpurun_out/profiles_out/<tag>_render.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r07}
FRAMES=5
mkdir -p "$ROOT/gpurun_out/profiles_out"
cd /tmp && export TMPDIR=/tmp
# the model bench.py renders: its default run up to the end of the timed region (same seed, pretrain count, steps)
python "$ROOT/bench.py" --no_cpu_baseline --no_seal --no_long_run --no_tensorf --no_render --save_model /tmp/s3d_model.pth > /tmp/render_train.log 2>&1 || { tail -5 /tmp/render_train.log; exit 1; }
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/render_prof -- python "$ROOT/tools/render_frames.py" --load /tmp/s3d_model.pth --frames $FRAMES > /tmp/render_prof.log 2>&1
LINE=$(grep "render 800x800" /tmp/render_prof.log | tail -1)
F=$(find /tmp/render_prof -name "*kernel_stats.csv" | head -1)
python - "$F" "$TAG" "$FRAMES" "$LINE" > "$ROOT/gpurun_out/profiles_out/${TAG}_render.md" <<'PY'
import csv, sys
f, tag, frames, line = sys.argv[1], sys.argv[2], int(sys.argv[3]) + 1, sys.argv[4]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 kernel trace `{tag}` — render-only workload (`tools/render_frames.py`, `tools/profile_render.sh`)\n")
print(f"{line}\n")
print(f"800x800 frame of the model `bench.py` renders (its default run: 384 pretraining + 16 warm-up + 64 timed steps, saved with `--save_model`), `NeRFRenderer.run_cuda` inference loop, "
      f"infer_batch_scale 4, sync_every 4; {frames} frames in the trace (one warm-up).  GPU kernel time per frame: {tot/frames/1e6:.2f} ms.\n")
print("| kernel | launches/frame | us/launch | ms/frame | % of kernel time |")
print("|---|---|---|---|---|")
for r in rows[:16]:
    n = r["Name"].replace("void ", "").replace("s3d::(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:80]
    print(f"| `{n}` | {int(r['Calls'])/frames:.1f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/frames/1e6:.3f} | {float(r['Percentage']):.1f} |")
PY
cat "$ROOT/gpurun_out/profiles_out/${TAG}_render.md"
