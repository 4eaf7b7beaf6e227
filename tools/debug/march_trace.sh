#!/bin/bash
# GPU box: per-dispatch durations of the inference-loop kernels over one frame (kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
python "$ROOT/tools/render_frames.py" --save /tmp/s3d_model.pth > /tmp/render_train.log 2>&1 || { tail -5 /tmp/render_train.log; exit 1; }
rm -rf /tmp/mt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -- python $ROOT/tools/render_frames.py --load /tmp/s3d_model.pth --frames 1 > /tmp/mt.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/mt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = ("k_march_rays", "k_composite_rays", "k_grid_forward_pair", "k_ffmlp_forward", "k_compact")
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows if any(n in r["Kernel_Name"] for n in names)]
# last frame only: the last 32 march launches
idx = [i for i, s in enumerate(seq) if "k_march_rays" in s[0]]
start = idx[-64] if len(idx) >= 64 else idx[0]
it = -1
for name, us, grid in seq[start:]:
    short = [n for n in names if n in name][0]
    if short == "k_march_rays":
        it += 1
        print(f"\niter {it:2d} grid={grid:>9s}:", end="")
    print(f" {short.replace('k_','')[:10]}={us:7.1f}", end="")
print()
PY
