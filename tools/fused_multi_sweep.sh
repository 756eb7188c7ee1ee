#!/bin/bash
# The one-kernel default call with several outputs beside the two-kernel route: n = 2^22 draws.
# Run on the GPU box:  bash tools/fused_multi_sweep.sh > gpurun_out/fused_both_multi.txt      (SPECS="k:l:kind ..." to choose)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in ${SPECS:-32:4:lin 16:4:lin 32:2:g 6:2:g 16:2:poly 8:8:lin 64:2:poly 6:5:g 3:8:g 12:5:lin 10:6:lin 4:8:lin 2:8:lin 8:4:g 4:4:g 8:3:poly 20:3:g 12:3:lin 16:3:lin 8:6:lin 9:7:g 5:6:poly 30:3:g 37:3:lin 22:3:poly 28:4:g}; do
    IFS=: read K L KIND <<< "$spec"
    K=$K L=$L KIND=$KIND LOG2N=22 python3 $R/tools/fused_both.py 2>&1 | grep -v checksum | tail -2 | cut -c1-200 || exit 1
done
