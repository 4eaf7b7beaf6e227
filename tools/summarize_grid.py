#!/usr/bin/env python3
"""gpurun_out/grid_<tag>/ (tools/profile_grid.sh) -> profiles/<tag>_grid_isolated.md: the isolated hash-grid encoder kernels
(Lego configuration, fp16) on random and ray-ordered points — time / points per second / fraction of the 8 TB/s roofline at 588
algorithmic bytes per point, and the rocprofv3 counters that show what bounds them."""
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = f"gpurun_out/grid_{tag}"
os.makedirs("profiles", exist_ok=True)
out = [f"# Isolated hash-grid encoder kernels `{tag}` — `python tools/bench_grid.py` (L16 F2 T2^19, fp16 tables, MI355X)\n",
       "588 algorithmic bytes per point (SURVEY §8d); `frac_hbm` = points/s x 588 B / 8 TB/s.  `random` = uniform points in the unit cube, "
       "`ray` = samples marched along 800x800 Lego-camera rays through the synthetic occupancy grid (the order training sees).\n"]
for f in sorted(glob.glob(f"{src}/timing_*.log")):
    out.append("```")
    out += [l.rstrip() for l in open(f) if l.startswith("grid_") or l.startswith("device")]
    out.append("```\n")
for f in sorted(glob.glob(f"{src}/stats_*.txt")):
    out.append("Kernel durations under `rocprofv3 --kernel-trace --stats` (B = 2^18, random and ray-ordered launches averaged):\n\n```")
    out += [l.rstrip() for l in open(f) if "k_grid" in l or "k_bin" in l]
    out.append("```\n")
for f in sorted(glob.glob(f"{src}/pmc_*.txt")):
    out.append("Counters per launch (separate `--pmc` passes, `--kernel-trace` only; averages over the launches of the kernel at B = 2^18):\n")
    cur, rows = None, {}
    for l in open(f):
        m = re.match(r"## (\S+)", l)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"(\S+)\s+([0-9.]+)\s+\(n=", l)
        if m and cur:
            rows.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    names = sorted({c for r in rows.values() for c in r})
    out.append("| counter | " + " | ".join(f"`{k}`" for k in rows) + " |")
    out.append("|---|" + "---|" * len(rows))
    for c in names:
        out.append(f"| {c} | " + " | ".join(f"{rows[k].get(c, float('nan')):.4g}" for k in rows) + " |")
    out.append("")
    for k, r in rows.items():
        if "TCC_HIT_sum" in r:
            out.append(f"* `{k}`: L2 hit rate {r['TCC_HIT_sum'] / max(r['TCC_HIT_sum'] + r.get('TCC_MISS_sum', 0), 1):.3f}; "
                       f"L2 requests per point {r.get('TCC_REQ_sum', 0) / 262144:.1f}; L1 (TCP) accesses per point "
                       f"{r.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / 262144:.1f}")
    out.append("")
open(f"profiles/{tag}_grid_isolated.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
