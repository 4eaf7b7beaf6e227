#!/bin/bash
# Build variants of libseal3d_hip.so that differ in the -D switches of ONE source file (run HERE, not on the GPU box):
#   tools/build_variants.sh gridencoder "base:" "nsub1:-DS3D_BIN3_NSUB=1" "p1024:-DS3D_BIN3_P=1024" ...
#     -> seal-3d_amd/csrc/build/variants/lib_<tag>.so      (travels to the GPU box with the snapshot)
# then on the box:  S3D_HIP_LIB=<variant.so> python tools/bench_grid.py ...
set -e
cd "$(dirname "$0")/../seal-3d_amd/csrc"
SRC=$1; shift
mkdir -p build/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -Wno-unused-variable"
for v in "$@"; do
  tag=${v%%:*}; defs=${v#*:}
  hipcc $FLAGS $defs -c $SRC.hip -o build/variants/${SRC}_$tag.o &
done
wait
OTHERS=$(ls build/*.o | grep -v "build/$SRC.o")
for v in "$@"; do
  tag=${v%%:*}
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$tag.so $OTHERS build/variants/${SRC}_$tag.o
  rm -f build/variants/${SRC}_$tag.o
done
ls -la build/variants/*.so
