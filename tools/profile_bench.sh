#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 passes over the default bench command.
#   1. --kernel-trace --stats           -> per-kernel durations
#   2. --pmc FETCH_SIZE                  -> HBM-side read KiB per dispatch   (own pass: TCC slots)
#   3. --pmc WRITE_SIZE                  -> HBM-side write KiB per dispatch  (own pass)
#   4. --pmc SQ_* (issue / wait mix)     -> where the dominant kernel's cycles go
#   5. --pmc MFMA counters               -> matrix-core utilisation of the ffmlp kernels
#   6. --pmc TCP / TCC counters          -> L1 accesses, L2 requests and hit rate of the grid kernels
# `profile_bench.sh <tag> trace` runs pass 1 only, `<tag> traffic` passes 1-3.
# PMC passes use --kernel-trace only (never sys/hip/hsa trace domains together with --pmc).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-r01}
ARGS="--steps 32 --warmup 8 --no_cpu_baseline --no_render --no_seal --no_long_run --no_tensorf"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
if [ "${2:-full}" != "trace" ]; then
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python "$ROOT/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python "$ROOT/bench.py" $ARGS > "$OUT/write.log" 2>&1
fi
if [ "${2:-full}" = "full" ]; then
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/sq" -- python "$ROOT/bench.py" $ARGS > "$OUT/sq.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python "$ROOT/bench.py" $ARGS > "$OUT/mfma.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/cache" -- python "$ROOT/bench.py" $ARGS > "$OUT/cache.log" 2>&1
fi
for f in trace fetch write sq mfma cache; do [ -f "$OUT/$f.log" ] && tail -1 "$OUT/$f.log" | cut -c1-400; done
# summarise on the box; only the summaries travel back (raw counter CSVs are hundreds of MB)
cd "$ROOT" && python tools/summarize_prof.py "${1:-r01}" > "$OUT/summary.log" 2>&1; tail -40 "$OUT/summary.log"
mkdir -p "$ROOT/gpurun_out/profiles_out" && cp "$ROOT"/profiles/${1:-r01}_* "$ROOT/gpurun_out/profiles_out/" 2>/dev/null
rm -rf "$OUT"
