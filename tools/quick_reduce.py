"""Quick on-GPU timing sweep of the eval and first-order reduction kernels (tuning aid)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native  # noqa: E402
from varsens_amd.saltelli import seed1_permutation  # noqa: E402

k = int(os.environ.get("K", 32))
n = 1 << int(os.environ.get("LOG2N", 22))
ctx = _native.Context(0)
print(ctx.info(), flush=True)
t0 = time.time()
perm = seed1_permutation(n)
print("perm %.3fs" % (time.time() - t0), flush=True)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm)
obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
N = 2 * (k + 1) * n
f = ctx.alloc(8 * N)
acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))


def timed(fn, reps=10):
    fn(); ctx.sync()
    ctx.profile_begin(1024)
    for _ in range(reps):
        fn()
    recs = ctx.profile_end()
    out = {}
    for name, ms in recs:
        out.setdefault(name, []).append(ms)
    return {kk: (np.median(v), np.min(v)) for kk, v in out.items()}


r = timed(lambda: ctx.eval(design, obj, 0, n, f), 5)
ms = r["eval"][0]
print("eval: %.3f ms  -> %.1f G evals/s, write %.1f GB/s" % (ms, N / ms / 1e6, 8 * N / ms / 1e6), flush=True)
def pipeline():
    ctx.eval(design, obj, 0, n, f)
    ctx.reduce_first_order(f, n, k, 1, acc)


for rev in (0, 1):
    os.environ["VS_REDUCE_REVERSE"] = str(rev)
    r = timed(pipeline, 10)
    print("pipeline reverse=%d: eval %.3f reduce %.3f (min %.3f) fold %.3f" %
          (rev, r["eval"][0], r["reduce_first"][0], r["reduce_first"][1], r["fold"][0]), flush=True)
    r = timed(lambda: ctx.reduce_first_order(f, n, k, 1, acc))
    print("reduce alone reverse=%d: %.3f (min %.3f)" % (rev, r["reduce_first"][0], r["reduce_first"][1]), flush=True)
os.environ["VS_REDUCE_REVERSE"] = "0"
for nv in (4,):
    for wpc in (1, 2, 3, 4):
        os.environ["VS_REDUCE_NV"] = str(nv)
        os.environ["VS_REDUCE_WGS_PER_CU"] = str(wpc)
        r = timed(lambda: ctx.reduce_first_order(f, n, k, 1, acc))
        ms = r["reduce_first"][0]
        print("reduce nv=%d wgs/cu=%2d: %.3f ms (min %.3f) -> %.0f GB/s = %.1f GSamples/s ; fold %.3f ms" %
              (nv, wpc, ms, r["reduce_first"][1], 8 * N / ms / 1e6, N / ms / 1e6, r["fold"][0]), flush=True)

K2 = k
gram = ctx.alloc(8 * 3 * K2 * K2)
for g in (1, 2, 3, 4):
    os.environ["VS_GRAM_WGS_PER_CU"] = str(g)
    r = timed(lambda: ctx.reduce_second_order(f, n, k, 1, gram), 5)
    print("gram wgs/cu=%d: %.3f ms fold %.3f" % (g, r["second_order"][0], r["second_fold"][0]), flush=True)
del os.environ["VS_GRAM_WGS_PER_CU"]
r = timed(lambda: ctx.reduce_second_order(f, n, k, 1, gram), 5)
ms = r["second_order"][0]
print("second order: %.3f ms (fold %.3f) -> %.1f GB/s algorithmic (16 k n bytes), %.2f TFLOP/s useful (3 k^2 fma/row)" %
      (ms, r["second_fold"][0], 16.0 * k * n / ms / 1e6, 2.0 * 3 * k * k * n / ms / 1e9), flush=True)

# multi-output reduction throughput (planar layout): l = 4 on a quarter of the rows = same bytes
l4, n4 = 4, n // 4
f4 = ctx.alloc(8 * 2 * (k + 1) * n4 * l4)
obj4 = ctx.objective_create(_native.OBJ_LINEAR, k, l4, np.tile(np.arange(1, k + 1, dtype=float), (l4, 1)) * np.arange(1, l4 + 1)[:, None])
perm4 = seed1_permutation(n4)
design4 = ctx.design_create(_native.GEN_HALTON, k, n4, 1 + 20 * k, 0, None, None, perm4)
acc4 = ctx.alloc(8 * _native.acc_len(k, l4, 1))
r = timed(lambda: ctx.eval(design4, obj4, 0, n4, f4), 5)
print("eval l=4 n/4: %.3f ms" % r["eval"][0], flush=True)
r = timed(lambda: ctx.reduce_first_order(f4, n4, k, l4, acc4))
ms = r["reduce_first"][0]
print("reduce l=4 n/4: %.3f ms -> %.0f GB/s" % (ms, 8.0 * 2 * (k + 1) * n4 * l4 / ms / 1e6), flush=True)

for w in (1, 2, 3):
    os.environ["VS_FUSED_WGS_PER_CU"] = str(w)
    r = timed(lambda: ctx.eval_reduce_first_order(design, obj, 0, n, acc), 10)
    print("fused wgs/cu=%d: %.3f ms (min %.3f) fold %.3f -> %.1f GSamples/s" %
          (w, r["fused"][0], r["fused"][1], r["fold"][0], N / r["fused"][0] / 1e6), flush=True)
