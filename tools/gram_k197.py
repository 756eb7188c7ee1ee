"""Second order beyond 128 index rows at the reference's own study shape, k = 197 (varsens/profile/parameters.py:86-110,
model.R:61-65): the [C; D] macro kernel (gram_macro_kernel + gram_convert_kernel), alone and with the first-order sums
riding on its diagonal pairs (vs_reduce_first_second_order).  N= rows (default 2^20; the study's largest is 20480)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = int(os.environ.get("K", 197)), int(os.environ.get("N", 1 << 20))
ctx = _native.Context(0)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
obj = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, np.linspace(0.0, 99.0, k))
f = ctx.alloc(8 * 2 * (k + 1) * n)
ctx.profile_begin(64)
for _ in range(4):
    ctx.eval(design, obj, 0, n, f)
ev = [ms for nm, ms in ctx.profile_end() if nm == "eval"]
gram = ctx.alloc(8 * _native.gram_len(k))
acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
reps = 10 if n <= (1 << 17) else 5
def timed(fn):
    for _ in range(2):
        fn()
    ctx.sync()
    ctx.profile_begin(256)
    for _ in range(reps):
        fn()
    recs = ctx.profile_end()
    per = {}
    for nm, ms in recs:
        per.setdefault(nm, []).append(ms)
    return {nm: float(np.sum(v)) / reps for nm, v in per.items()}
so = timed(lambda: ctx.reduce_second_order(f, n, k, 1, gram))
g2 = ctx.download(gram, (2, k, k))
fo = timed(lambda: ctx.reduce_first_order(f, n, k, 1, acc))
bo = timed(lambda: ctx.reduce_first_second_order(f, n, k, 1, acc, gram))
g = ctx.download(gram, (2, k, k))
nb = (2 * k + 127) // 128
tiles = nb * 36 + nb * (nb - 1) // 2 * 64
flops = tiles * (n / 4.0) * (2 * 16 * 16 * 4)
ms = so.get("second_order", float("nan"))
print("k=%d n=%d: evaluation %.3f ms (%.2f TB/s written)" % (k, n, np.median(ev), 8.0 * 2 * (k + 1) * n / np.median(ev) / 1e9))
print("second order alone: %s  -> Gram kernel %.3f ms = %.1f TFLOP/s issued (%d tiles of 16x16 per 4 rows), %.2f TB/s of operands"
      % (so, ms, flops / ms / 1e9, tiles, 16.0 * k * n / ms / 1e9))
print("first order alone: %s" % fo)
print("both orders in one call: %s  -> %.3f ms of kernels" % (bo, sum(bo.values())))
print("max |SS, TT (one call) - (second order alone)| / max|.| = %.2e" % (np.abs(g - g2).max() / np.abs(g2).max()))
