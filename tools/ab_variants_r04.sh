R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/build_variants/libvarsens_hip_slds2.so; do
  echo "== lib: ${lib:-product}"
  for kind in lin g poly; do VARSENS_HIP_LIB=$lib python3 $R/tools/eval_one.py 32 22 fused $kind; done
  VARSENS_HIP_LIB=$lib python3 $R/tools/eval_one.py 25 22 fused lin
done
for lib in "" $R/build_variants/libvarsens_hip_gs2.so; do
  echo "== lib: ${lib:-product}"
  VARSENS_HIP_LIB=$lib K=128 LOG2N=24 python3 $R/tools/gram_only.py | tail -2
  VARSENS_HIP_LIB=$lib K=100 LOG2N=22 python3 $R/tools/gram_only.py | tail -2
done
