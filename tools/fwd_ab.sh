cd $GRAFT_REPO_ROOT
V=seal-3d_amd/csrc/build/variants
for v in default nobal r50 r120; do
  if [ $v = default ]; then unset S3D_HIP_LIB; else export S3D_HIP_LIB=$GRAFT_REPO_ROOT/$V/lib_$v.so; fi
  echo "== $v"; timeout 120 python tools/bench_grid.py --no_bwd --sum --iters 60 --sizes 262144 2097152 2>&1 | grep grid_fwd
done
unset S3D_HIP_LIB
echo "== levels"; timeout 200 python tools/fwd_levels.py 2>&1 | tail -20
