for v in base a64x2t512 a64x2t1024 base; do
  echo "== $v"; S3D_HIP_LIB=seal-3d_amd/csrc/build/variants/lib_$v.so timeout 300 python tools/bench_grid_adam.py 2>&1 | tail -2
done
