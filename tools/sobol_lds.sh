#!/bin/bash
# Sobol designs through the scan kernels, nibble tables in LDS (round 5) against the byte tables in memory (round 4) and
# against Halton designs through the specialised kernels and through the same general column code.
#   here:        python -m varsens_amd.build --variant bytetab -DVS_SOBOL_BYTE_TABLES      (the A/B library)
#   on the GPU:  bash tools/sobol_lds.sh > gpurun_out/r05/sobol_lds.txt
cd "$(dirname "$0")/.."
AB=$PWD/build_variants/libvarsens_hip_bytetab.so
echo "tools/sobol_lds.sh: linear objective, identity scaling, warm launches, medians (ms); library build $(python -c 'from varsens_amd import _native; print(_native.build_id())')"
for KN in "16 22" "32 22" "64 21" "96 20" "128 20"; do
    set -- $KN
    echo "--- k = $1, n = 2^$2"
    K=$1 LOG2N=$2 python tools/sobol_vs_halton.py | sed 's/^/                         /'
    [ -f "$AB" ] && VARSENS_HIP_LIB=$AB K=$1 LOG2N=$2 python tools/sobol_vs_halton.py | grep sobol | sed 's/^/byte tables in memory:   /'
    VS_TUNING=1 VS_SCAN_GENERAL=1 K=$1 LOG2N=$2 python tools/sobol_vs_halton.py | grep halton | sed 's/^/general column code:    /'
done
