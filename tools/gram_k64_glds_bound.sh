#!/bin/bash
# VERDICT r4 item 3, the "one measured attempt" at operands straight into LDS for the k <= 64 second-order kernel:
# product against build_variants/libvarsens_hip_glds.so (python -m varsens_amd.build --variant glds -DVS_ST_GLDS_TIMING):
# gram_st_kernel with its eight global_load_dwordx4 + sixteen ds_write_b128 per thread and stage replaced by eight
# global_load_lds_dwordx4 (lane-linear 1 KiB pieces, nothing forms s and t: WRONG RESULTS, timing only) -- the most a
# kernel built around LDS-DMA could gain on the operand path, before it pays for forming s and t at fragment-read time.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rnd in 1 2 3; do
for lib in "" $R/build_variants/libvarsens_hip_glds.so; do
  for K in 64; do   # (the experiment's stage layout is the 64-row one)
    echo "lib ${lib:-product}: $(VARSENS_HIP_LIB=$lib K=$K LOG2N=22 python3 $R/tools/gram_only.py | tail -1)"
  done
done
done
