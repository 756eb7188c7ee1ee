R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/build_variants/libvarsens_hip_occ.so; do
  echo "== lib: ${lib:-product}"
  VARSENS_HIP_LIB=$lib SHAPES=16x2,8x4,6x2,4x3,10x3,32x1 LOG2N=22 python3 $R/tools/gram_multi.py | cut -c1-80
  VARSENS_HIP_LIB=$lib SHAPES=16x2,6x2,32x1 LOG2N=20 python3 $R/tools/gram_multi.py | cut -c1-80
done
