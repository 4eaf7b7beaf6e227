for v in "" tiny0 tiny16; do
 lib=""; [ -n "$v" ] && lib=$PWD/seal-3d_amd/csrc/build/variants/lib_$v.so
 echo "== $v"; S3D_HIP_LIB=$lib bash tools/profile_tensorf.sh r11 quick 2>&1 | grep -E "graph trainer|k_vm_plane|k_vm_line"
done
