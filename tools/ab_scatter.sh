#!/bin/bash
# GPU box: scatter knobs of the binned grid backward (tools/build_variants.sh gridencoder ...): bench A/B
for v in "$@"; do
  echo "== lib_$v"
  S3D_HIP_LIB=seal-3d_amd/csrc/build/variants/lib_$v.so python bench.py --steps 32 --warmup 8 --no_long_run --no_cpu_baseline --no_render --no_seal --no_tensorf 2>/dev/null | python -c "
import json,sys
s=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=s['roofline']
print(round(s['value']/1e6,1), round(s['ms_per_step'],4), round(r['frac'],4), round(r['avg_us'],1), round(r['grid_backward_plain']['avg_us'],1))"
done
