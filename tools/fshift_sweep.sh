#!/bin/bash
# bench.py and the config-5 kernels by rows per row block of the f-value layout (VS_F_SHIFT; 40 = planar)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rnd in 1 2; do
for sh in 40 12 11 13 14; do
  VS_TUNING=1 VS_F_SHIFT=$sh python3 $R/bench.py --no-host-routes --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python3 - $sh <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print("shift %s: value %.1f  step %.4f  eval %.4f  reduce_first %.4f (%.3f)  gram k32 %.4f  both-orders call %.4f  api two-kernel %.4f" % (sys.argv[1], d["value"], d["ms_per_step"], d["stages_ms"]["eval"], d["stages_ms"]["reduce_first"], d["roofline"]["frac"], d["roofline_second_order"]["avg_launch_ms"], d["both_orders_one_call"]["ms_per_call"], d["api_default"]["two_kernel_path"]["ms_per_pass"]), flush=True)
PY
done
done
for sh in 40 12 11 13; do
  echo "config 5, shift $sh"; VS_TUNING=1 VS_F_SHIFT=$sh python3 $R/tools/config5.py 2>&1 | sed -n 2,3p
done
