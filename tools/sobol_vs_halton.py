"""Sobol vs Halton designs through the same kernels (k = K, n = 2^LOG2N, linear objective): evaluation, fused first order,
the one-kernel default call -- medians of warm launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation, sobol_directions
k, n = int(os.environ.get("K", 32)), 1 << int(os.environ.get("LOG2N", 22))
ctx = _native.Context(0)
perm = seed1_permutation(n)
designs = {"halton": ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm),
           "sobol": ctx.design_create(_native.GEN_SOBOL, k, n, 4096, 0, None, None, perm, None, sobol_directions(k))}
obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
f = ctx.alloc(8 * 2 * (k + 1) * n)
acc, gram = ctx.alloc(8 * _native.acc_len(k, 1, 1)), ctx.alloc(8 * _native.gram_len(k))
for name, d in designs.items():
    res = {}
    for what, fn, key in (("eval", lambda: ctx.eval(d, obj, 0, n, f), "eval"),
                          ("fused first order", lambda: ctx.eval_reduce_first_order(d, obj, 0, n, acc), "fused"),
                          ("one-kernel default call", lambda: ctx.eval_reduce_first_second_order(d, obj, 0, n, acc, gram), "fused")):
        for _ in range(30):
            fn()
        ctx.sync()
        ctx.profile_begin(512)
        for _ in range(20):
            fn()
        res[what] = float(np.median([ms for nm, ms in ctx.profile_end() if nm == key and ms > 0.02]))
    print("%-7s k=%d n=2^%d: " % (name, k, int(np.log2(n))) + "   ".join("%s %.3f ms" % kv for kv in res.items()), flush=True)
