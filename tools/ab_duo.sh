set -e
python -m pytest tests/test_gpu_ffmlp.py -q -x 2>&1 | tail -2
export S3D_BENCH_SIZES=269824
for mix in 4x4 6x2; do
  echo "== mix $mix"; S3D_DUO_MIX=$mix S3D_DUO_MIX_LIGHT=$mix python tools/bench_ffmlp.py 2>&1 | grep "B=" | sed 's/fwd(train).*bwd(2-kernel)/bwd(2-kernel)/'
done
