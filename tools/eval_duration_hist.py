"""Histogram of the per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV:
    python3 tools/eval_duration_hist.py TRACE.csv 'scan_kernel<3, 8, 4, false' [min_us]"""
import csv, sys
import numpy as np
path, pat = sys.argv[1], sys.argv[2]
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
d = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        if pat in r["Kernel_Name"]:
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if us >= lo:
                d.append(us)
d = np.array(d)
print("%d launches >= %.0f us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" %
      (len(d), lo, d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
h, e = np.histogram(d, bins=12)
for c, a, b in zip(h, e[:-1], e[1:]):
    print("  %7.1f .. %7.1f us  %4d %s" % (a, b, c, "#" * int(60 * c / max(1, h.max()))))
print("in launch order (us, every 8th):", " ".join("%.0f" % x for x in d[::8][:60]))
# by position in a rotation of R buffers (R=3: bench.py's f-value buffers): does one of them cost more?
R = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for m in range(R):
    print("  launches %d mod %d: mean %.1f us  median %.1f  (%d)" % (m, R, d[m::R].mean(), np.median(d[m::R]), len(d[m::R])))
