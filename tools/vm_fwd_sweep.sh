ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for t in y0 yg yb; do
  rm -rf /tmp/vm_sweep
  S3D_HIP_LIB=$ROOT/seal-3d_amd/csrc/build/variants/lib_$t.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vm_sweep -- python "$ROOT/tools/bench_tensorf_step.py" 300 fused native > /tmp/vm_sweep.log 2>&1
  F=$(find /tmp/vm_sweep -name "*kernel_stats.csv" | head -1)
  echo "variant $t: $(grep -h 'k_vm_color_basis\|k_vm_features' $F | awk -F, '{print $1" "$4}' | sed 's/s3d::(anonymous namespace):://; s/(.*)//' | tr '\n' ' ')"
done
