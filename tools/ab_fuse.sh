ARGS="--steps 40 --warmup 10 --no_cpu_baseline --no_render --no_seal --no_long_run --no_tensorf"
V=$PWD/seal-3d_amd/csrc/build/variants
for rep in 1 2; do
for cfg in "1:" "0:" "1:$V/lib_acc64x2.so" "0:$V/lib_acc64x2.so"; do
  f=${cfg%%:*}; lib=${cfg#*:}
  S3D_HIP_LIB=$lib S3D_FUSE_TABLE_ADAM=$f python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse=$f lib=$(basename "$lib" 2>/dev/null)', 'ms/step', round(d['ms_per_step'],4), 'samples/s %.3e' % d['value'], 'samples/step', int(d['config']['samples_per_step']))"
done; done
