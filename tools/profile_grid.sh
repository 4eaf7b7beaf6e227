#!/bin/bash
# Runs on the GPU box (via gpurun): isolated hash-grid encoder kernels (tools/bench_grid.py), timing + rocprofv3 passes.
#   profile_grid.sh <tag> [fwd_variant_list] [bwd_variant_list]
# Writes gpurun_out/grid_<tag>/{timing_*.log, stats_*.txt, pmc_*.txt}; PMC passes use --kernel-trace only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
FV=${2:-"0"}   # (forward / backward variant switches of past experiments; the library has none now)
BV=${3:-"0"}
OUT=$ROOT/gpurun_out/grid_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for fv in $FV; do for bv in $BV; do
  export S3D_GRID_FWD=$fv S3D_GRID_BWD=$bv
  v=f${fv}b${bv}
  timeout 300 python "$ROOT/tools/bench_grid.py" --iters 20 --sum > "$OUT/timing_$v.log" 2>&1
  cat "$OUT/timing_$v.log"
  P="$OUT/prof_$v"
  ARGS="--iters 4 --sizes 262144"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python "$ROOT/tools/bench_grid.py" $ARGS > "$P.trace.log" 2>&1
  python "$ROOT/tools/kstats.py" "$P/trace" 1.0 > "$OUT/stats_$v.txt" 2>&1
  i=0
  for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES" \
             ${S3D_PMC_EXTRA:-}; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$P/pmc$i" -- python "$ROOT/tools/bench_grid.py" $ARGS > "$P.pmc$i.log" 2>&1 || echo "pmc pass $i failed" >> "$OUT/pmc_$v.txt"
    for k in k_grid_forward k_bin_scatter k_bin_accumulate; do
      if ls "$P/pmc$i" >/dev/null 2>&1; then echo "## $k  [$set]" >> "$OUT/pmc_$v.txt"; python "$ROOT/tools/pmc_kernel.py" "$P/pmc$i" $k >> "$OUT/pmc_$v.txt" 2>/dev/null; fi
    done
  done
  tail -3 "$P.pmc1.log" | cut -c1-300
  rm -rf "$P"
done; done
cd "$ROOT" && python tools/summarize_grid.py "$TAG" > "$OUT/summary.log" 2>&1; tail -60 "$OUT/summary.log"
mkdir -p "$ROOT/gpurun_out/profiles_out" && cp "$ROOT"/profiles/${TAG}_grid_isolated.md "$ROOT/gpurun_out/profiles_out/" 2>/dev/null
