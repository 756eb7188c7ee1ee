#!/bin/bash
# The one-kernel default call with several outputs beside the two-kernel route: n = 2^22 draws.
# Run on the GPU box:  bash tools/fused_multi_sweep.sh > gpurun_out/fused_both_multi.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in ${SPECS:-"32 4 lin" "16 4 lin" "32 2 g" "6 2 g" "16 2 poly" "8 8 lin" "64 2 poly" "6 5 g" "3 8 g" "12 5 lin" "10 6 lin" "4 8 lin" "2 8 lin" "8 4 g" "4 4 g" "8 3 poly" "20 3 g" "12 3 lin" "16 3 lin" "8 6 lin" "9 7 g" "5 6 poly"}; do
    set -- $spec
    K=$1 L=$2 KIND=$3 LOG2N=22 python3 $R/tools/fused_both.py | grep -v checksum || exit 1
done
