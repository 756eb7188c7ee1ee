"""Does the HBM stream cost the matrix pipe its throughput?    The SAME launch of the second-order
kernel -- k = K index rows per half, n = 2^LOG2N draws, small enough for the f-values to fit the 256 MB Infinity
Cache -- timed two ways:
  MODE=cache  the same buffer every launch (after the first launch the values come from the memory-side cache);
  MODE=hbm    BUFS different buffers in rotation, together far beyond the cache: every launch streams from HBM.
Same grid, same prologue and tail, same arithmetic; only where the bytes come from differs.

    K=64 LOG2N=17 MODE=cache python3 tools/gram_cache_vs_hbm.py
    K=64 LOG2N=17 MODE=hbm python3 tools/pmc_run.py gram_k64_hbm --kernels gram_ -- python3 tools/gram_cache_vs_hbm.py
(tools/pmc_run.py adds the clock -- GRBM_GUI_ACTIVE / 8 / duration -- and the matrix pipe's busy share.)
WHAT=both times vs_reduce_first_second_order instead (first-order sums on the same fragments).
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation

k, log2n = int(os.environ.get("K", 64)), int(os.environ.get("LOG2N", 17))
mode, what = os.environ.get("MODE", "cache"), os.environ.get("WHAT", "second")
n = 1 << log2n
bytes_f = 8 * 2 * (k + 1) * n
bufs = 1 if mode == "cache" else int(os.environ.get("BUFS", max(2, int(6e9 // bytes_f))))
ctx = _native.Context(0)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
fs = [ctx.alloc(bytes_f) for _ in range(bufs)]
for f in fs:
    ctx.eval(design, obj, 0, n, f)
gram = ctx.alloc(8 * _native.gram_len(k))
acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
reps = int(os.environ.get("REPS", 200))


def launch(i):
    f = fs[i % bufs]
    if what == "both":
        ctx.reduce_first_second_order(f, n, k, 1, acc, gram)
    else:
        ctx.reduce_second_order(f, n, k, 1, gram)


for i in range(max(bufs, 20)):
    launch(i)
ctx.sync()
ctx.profile_begin(8 * reps + 8)
for i in range(reps):
    launch(i)
t = [ms for nm, ms in ctx.profile_end() if nm == "second_order"]
med = float(np.median(t))
nbl = (k + 15) // 16
flops = nbl * (nbl + 1) * (n / 4.0) * 2048
print("k=%d n=2^%d %s mode=%s buffers=%d (%.0f MB each): %.4f ms per launch (min %.4f, p90 %.4f) = %.2f TB/s read, "
      "%.1f TFLOP/s fp64 MFMA; scaled to n=2^22: %.3f ms" %
      (k, log2n, what, mode, bufs, bytes_f / 1e6, med, min(t), float(np.percentile(t, 90)),
       16.0 * k * n / med / 1e9, flops / med / 1e9, med * (1 << 22) / n), flush=True)
