#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcg
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/gram_only.py > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(list)
for f in glob.glob(R+"/gpurun_out/pmcg/p*/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        if "gram_diag_wave_kernel" in row["Kernel_Name"] or "gram_offdiag_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    v=agg[k]; print("%-28s %14.0f (n=%d)" % (k,sum(v)/len(v),len(v)))
PY
