#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC traffic passes.
# Summaries land in gpurun_out/profiles/ ; copy the ones to be judged into profiles/rNN/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json, os, shutil
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/profiles"
stats = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], out + "/bench_kernel_stats.csv")
pmc = {}
for which in ("fetch", "write"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % which, recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (kname, c), v in agg.items():
            pmc.setdefault(kname, {})[c + "_KB_avg_per_launch"] = sum(v) / len(v)
            pmc[kname]["launches_" + c] = len(v)
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 1`; "
                   "raw counter units are KB; on gfx950 FETCH_SIZE counts half the bytes of 16-B-per-lane coalesced "
                   "loads (MI355X_MICROARCH.md, HBM) -- doubled by the reader, not here",
           "kernels": pmc}, open(out + "/bench_pmc_fetch_write.json", "w"), indent=1)
for d in ("stats", "fetch", "write"):
    shutil.rmtree(out + "/" + d, ignore_errors=True)
PY
cat $OUT/bench_n1.json
cat $OUT/bench_kernel_stats.csv
