#!/bin/bash
# PMC passes over one shape of the scan kernels: bash tools/pmc_scan.sh K LOG2N eval|fused
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_scan
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/eval_one.py $1 $2 $3 > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
agg=collections.defaultdict(list)
for f in glob.glob(R+"/gpurun_out/pmc_scan/p*/**/*counter_collection.csv",recursive=True):
    rows=list(csv.DictReader(open(f)))
    big=max(int(r["Grid_Size"]) for r in rows if "scan_kernel" in r["Kernel_Name"])
    for row in rows:
        if "scan_kernel" in row["Kernel_Name"] and int(row["Grid_Size"])==big:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
w=sum(agg["SQ_WAVES"])/len(agg["SQ_WAVES"])
for k in sorted(agg):
    v=sum(agg[k])/len(agg[k]); print("%-24s %16.0f   per wave %10.1f" % (k,v,v/w))
PY
