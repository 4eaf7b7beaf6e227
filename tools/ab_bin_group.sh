#!/bin/bash
# GPU box: interleave group of the binned grid backward (S3D_BIN_GROUP_LOG variants from tools/build_variants.sh): tests + bench A/B
for g in g10; do
  echo "== tests with lib_$g"; S3D_HIP_LIB=seal-3d_amd/csrc/build/variants/lib_$g.so timeout 600 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_optim.py -x -q 2>&1 | tail -1
done
for v in g5 g8 g9 g10 g11 g5 g9 g10; do
  echo "== lib_$v"
  S3D_HIP_LIB=seal-3d_amd/csrc/build/variants/lib_$v.so python bench.py --steps 32 --warmup 8 --no_long_run --no_cpu_baseline --no_render --no_seal --no_tensorf 2>/dev/null | python -c "
import json,sys
s=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=s['roofline']
print(round(s['value']/1e6,1), round(s['ms_per_step'],4), round(r['frac'],4), round(r['avg_us'],1), round(r['grid_backward_plain']['avg_us'],1))"
done
