#!/bin/bash
# GPU box: tools/vm_variant_sweep.sh for each variant, once with the 32-points-per-trip kernels (S3D_VM_MM=1) and once without
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for mm in 0 1; do
  echo "#### S3D_VM_MM=$mm"
  S3D_VM_MM=$mm bash "$ROOT/tools/vm_variant_sweep.sh" "$@"
done
