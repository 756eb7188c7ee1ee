#!/bin/bash
# run the eval->reduce pipeline timing for each build variant (tuning aid)
for v in "" $(ls build_variants/*.so); do
  echo "== variant: ${v:-default}"
  VARSENS_HIP_LIB=$v timeout -k 10 200 python tools/quick_reduce.py 2>&1 | grep -E "pipeline reverse=0|^eval"
done
