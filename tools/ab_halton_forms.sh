#!/bin/bash
# A/B of the Halton digit step in the straight-line scan kernels: product (three instructions per digit,
# vs_radical_inverse_pair_horner) vs build_variants/libvarsens_hip_digits.so (VS_EXTRA_CFLAGS=-DVS_HALTON_DIGITS: four).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/build_variants/libvarsens_hip_digits.so "" $R/build_variants/libvarsens_hip_digits.so; do
  echo "== lib: ${lib:-product}"
  for spec in "32 22 fused lin" "32 22 eval lin" "32 22 fused g" "64 21 fused lin" "128 20 fused poly" "6 22 fused g" "128 22 eval poly"; do
    VARSENS_HIP_LIB=$lib python3 $R/tools/eval_one.py $spec | tail -1
  done
  VARSENS_HIP_LIB=$lib K=32 LOG2N=22 KIND=lin python3 $R/tools/fused_both.py | grep "fused both"
  VARSENS_HIP_LIB=$lib K=128 LOG2N=22 KIND=poly python3 $R/tools/fused_both.py | grep "fused both"
done
