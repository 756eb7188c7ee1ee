#!/bin/bash
# PMC passes for the eval/reduce kernels (diagnostic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(list)
for f in glob.glob(R+"/gpurun_out/pmc/p*/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        agg[(row["Kernel_Name"][:30],row["Counter_Name"])].append(float(row["Counter_Value"]))
for k in sorted(agg):
    v=agg[k]; print("%-32s %-24s %14.0f" % (k[0],k[1],sum(v)/len(v)))
PY
