#!/bin/bash
# Run on the GPU box (via gpurun): the committed profiles/rNN set.
#   bench line; rocprofv3 --kernel-trace --stats of bench.py; counter passes (tools/pmc_run.py: one group per pass,
#   never together with sys/hip/hsa traces) over bench.py, over the Gram kernels at k = 32 / 64 / 128 and over the
#   config-5 evaluation and over the fused evaluate + both-orders kernels (k = 32, k = 128); BASELINE configs[4] on one GPU (tools/config5.py) with its kernel stats.
# Summaries land in gpurun_out/profiles/ ; copy the ones to be judged into profiles/rNN/.
# Every profiled program follows `--` itself (no env/bash hop: the profiler's preload initialises the GPU).
# Legs: `collect_profiles.sh` runs all of them; `collect_profiles.sh bench` only the bench line with its stats and counter
# passes (what bench.py's `roofline.traffic` stamp needs after a late source change), `kernels`, `config5`, `extras` likewise.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
LEGS=${*:-bench kernels config5 extras}
leg() { case " $LEGS " in *" $1 "*) return 0;; *) return 1;; esac; }
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
if leg bench; then
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done"
# (the host-route legs launch the same kernels at other sizes: kept out of the counter and trace passes)
python3 $R/tools/pmc_run.py bench_pmc -- python3 $R/bench.py --steps 5 --warmup 1 --prewarm-steps 20 --no-cpu-baseline --no-host-routes
cp $R/gpurun_out/pmc/bench_pmc.json $OUT/bench_pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-host-routes > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
echo "bench stats + pmc done"
fi
if leg kernels; then
K=128 LOG2N=24 python3 $R/tools/pmc_run.py gram_k128_n24 --kernels gram_,first_head,reduce_first -- python3 $R/tools/gram_only.py
K=64 LOG2N=22 python3 $R/tools/pmc_run.py gram_k64_n22 --kernels gram_,first_head,reduce_first -- python3 $R/tools/gram_only.py
K=32 LOG2N=22 python3 $R/tools/pmc_run.py gram_k32_n22 --kernels gram_,first_head,reduce_first -- python3 $R/tools/gram_only.py
python3 $R/tools/pmc_run.py eval_k128_n24 --kernels scan_kernel -- python3 $R/tools/eval_one.py 128 24 eval poly
K=32 LOG2N=22 KIND=lin python3 $R/tools/pmc_run.py fused_both_k32 --kernels scan_gram --groups issue,lds -- python3 $R/tools/fused_both.py
K=128 LOG2N=24 KIND=poly python3 $R/tools/pmc_run.py fused_both_k128_n24 --kernels scan_gram --groups issue,lds -- python3 $R/tools/fused_both.py
# several outputs in the one-kernel call (round 4): k = 32 x 4 outputs (eight waves), k = 16 x 2 (four waves, 32 index rows)
K=32 L=4 LOG2N=22 KIND=lin python3 $R/tools/pmc_run.py fused_both_k32_l4 --kernels scan_gram --groups issue,lds -- python3 $R/tools/fused_both.py
K=16 L=2 LOG2N=22 KIND=poly python3 $R/tools/pmc_run.py fused_both_k16_l2 --kernels scan_gram --groups issue,lds -- python3 $R/tools/fused_both.py
python3 $R/tools/pmc_run.py fused_first_k32 --kernels scan_kernel --groups issue,lds,mem -- python3 $R/tools/eval_one.py 32 22 fused lin
# round 5: second order beyond 128 index rows at the reference's study shape (k = 197, n = 2^20)
N=1048576 python3 $R/tools/pmc_run.py gram_k197_n2pow20 --kernels gram_,first_head,reduce_first,scan_kernel -- python3 $R/tools/gram_k197.py
# round 5: a Sobol design through the fused first-order kernel (points from nibble tables in LDS), k = 64 and 128
python3 $R/tools/pmc_run.py fused_first_k64_sobol --kernels scan_kernel --groups issue,lds -- python3 $R/tools/eval_one.py 64 21 fused lin sobol
python3 $R/tools/pmc_run.py fused_first_k128_sobol --kernels scan_kernel --groups issue,lds -- python3 $R/tools/eval_one.py 128 20 fused lin sobol
for n in gram_k128_n24 gram_k64_n22 gram_k32_n22 eval_k128_n24 fused_both_k32 fused_both_k128_n24 fused_both_k32_l4 fused_both_k16_l2 fused_first_k32 gram_k197_n2pow20 fused_first_k64_sobol fused_first_k128_sobol; do cp $R/gpurun_out/pmc/$n.json $R/gpurun_out/pmc/${n}_kernel_stats.csv $OUT/; done
echo "kernel pmc done"
fi
if leg config5; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $R/tools/config5.py > $OUT/config5.log 2>&1
cp $(find $OUT/stats5 -name '*kernel_stats.csv' | head -1) $OUT/config5_kernel_stats.csv
rm -rf $OUT/stats $OUT/stats5
echo "config5 done"
fi
if leg extras; then
# round 5: the routes with an opaque model (whole API calls by the wall clock), the row-block size of the f-value layout,
# and the one-GPU rehearsal of ranks 0, 3, 7 of an 8-GPU weak-scaling job
python3 $R/tools/host_routes.py 32 20 1 32 20 3 > $OUT/host_routes.txt 2>&1
bash $R/tools/fshift_sweep.sh > $OUT/fshift_sweep.txt 2>&1
python3 $R/tools/rehearse_weak_scaling.py --out $OUT/rehearsal > $OUT/rehearsal.log 2>&1
N=20480 python3 $R/tools/gram_k197.py > $OUT/gram_k197_n20480.txt 2>&1
python3 $R/tools/gram_k197.py > $OUT/gram_k197_n2pow20.txt 2>&1
bash $R/tools/sobol_lds.sh > $OUT/sobol_lds.txt 2>&1
python3 $R/tools/text_io_rate.py 32 14 > $OUT/text_io_rate.txt 2>&1
K=64,56,48,40 python3 $R/tools/gram_dma_ab.py > $OUT/gram_dma_ab_final.txt 2>&1
echo "round-5 extras done"
fi
for f in bench_n1.json bench_kernel_stats.csv config5.log config5_kernel_stats.csv; do [ -f $OUT/$f ] && cat $OUT/$f; done
