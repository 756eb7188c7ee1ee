#!/bin/bash
# does the HBM stream cost the matrix pipe its throughput?  tools/gram_cache_vs_hbm.py
#  (1) the product library, small designs, same buffer every launch vs buffers in rotation;
#  (2) a diagnostic build whose workgroups walk their stages 32 times (build_variants/libvarsens_hip_strep32.so:
#      hipcc ... -DVS_ST_REPEAT=32 vs_gram_st.hip, linked with the product's other objects): a 136 MB design -- inside the
#      256 MB Infinity Cache -- gives a launch as long as n = 2^22 from HBM;  with the counter passes for clock and pipe.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "64 17" "48 17" "128 16" "32 18"; do
    set -- $spec
    for mode in cache hbm; do
        K=$1 LOG2N=$2 MODE=$mode python3 $R/tools/gram_cache_vs_hbm.py || exit 1
    done
done
echo "--- product library, n = 2^22 from HBM"
K=64 LOG2N=22 MODE=cache REPS=40 python3 $R/tools/gram_cache_vs_hbm.py
K=128 LOG2N=21 MODE=cache REPS=40 python3 $R/tools/gram_cache_vs_hbm.py
if [ -f $R/build_variants/libvarsens_hip_strep32.so ]; then
    export VARSENS_HIP_LIB=$R/build_variants/libvarsens_hip_strep32.so
    echo "--- diagnostic build: 32 walks over the same stages (times are per launch of 32 walks)"
    K=64 LOG2N=17 MODE=cache REPS=40 python3 $R/tools/gram_cache_vs_hbm.py
    K=64 LOG2N=17 MODE=hbm REPS=40 python3 $R/tools/gram_cache_vs_hbm.py
    K=64 LOG2N=15 MODE=cache REPS=40 python3 $R/tools/gram_cache_vs_hbm.py
    K=128 LOG2N=16 MODE=cache REPS=40 python3 $R/tools/gram_cache_vs_hbm.py
    K=64 LOG2N=17 MODE=cache REPS=20 python3 $R/tools/pmc_run.py gram_k64_n17_x32_cache --kernels gram_st --groups issue,lds,fetch -- python3 $R/tools/gram_cache_vs_hbm.py
    K=64 LOG2N=15 MODE=cache REPS=20 python3 $R/tools/pmc_run.py gram_k64_n15_x32_cache --kernels gram_st --groups issue,lds,fetch -- python3 $R/tools/gram_cache_vs_hbm.py
    unset VARSENS_HIP_LIB
fi
K=64 LOG2N=22 MODE=cache REPS=20 python3 $R/tools/pmc_run.py gram_k64_n22_hbm --kernels gram_st --groups issue,lds,fetch -- python3 $R/tools/gram_cache_vs_hbm.py
