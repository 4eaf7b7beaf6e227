#!/bin/bash
# Build variants of libseal3d_hip.so with different k_bin_accumulate tuning knobs (run HERE, not on the GPU box):
#   tools/tune_bin.sh "64 1024 4" "64 512 8" ...       -> seal-3d_amd/csrc/build/variants/lib_<kb>_<threads>_<unroll>.so
# then on the box:  S3D_HIP_LIB=<variant.so> python tools/bench_grid_bwd.py 2 rays4096 f16
set -e
cd "$(dirname "$0")/../seal-3d_amd/csrc"
rm -rf build/variants; mkdir -p build/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
for v in "$@"; do
  set -- $v
  tag="$1_$2_$3"
  hipcc $FLAGS -DS3D_BIN_ACC_KB=$1 -DS3D_BIN_ACC_THREADS=$2 -DS3D_BIN_ACC_UNROLL=$3 -c gridencoder.hip -o build/variants/grid_$tag.o &
done
wait
for o in build/variants/grid_*.o; do
  tag=${o#build/variants/grid_}; tag=${tag%.o}
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$tag.so build/api.o build/encoders.o build/ffmlp.o build/raymarching.o $o
done
ls build/variants/*.so
