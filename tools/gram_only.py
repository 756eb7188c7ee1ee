"""Run only the second-order Gram kernel a few times and print its median time (also for rocprofv3 passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = int(os.environ.get("K", 32)), 1 << int(os.environ.get("LOG2N", 22))
ctx = _native.Context(0)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
f = ctx.alloc(8 * 2 * (k + 1) * n)
ctx.eval(design, obj, 0, n, f)
gram = ctx.alloc(8 * _native.gram_len(k))
for _ in range(3):
    ctx.reduce_second_order(f, n, k, 1, gram)
ctx.sync()
ctx.profile_begin(64)
for _ in range(10):
    ctx.reduce_second_order(f, n, k, 1, gram)
r = [ms for nm, ms in ctx.profile_end() if nm == "second_order"]
g = ctx.download(gram, (2, k, k))
print("gram checksum %.17g  SS[3,5]=%.17g" % (float(g.sum()), float(g[0, 3, 5])))
acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
for _ in range(3):
    ctx.reduce_first_order(f, n, k, 1, acc)
ctx.sync()
ctx.profile_begin(64)
for _ in range(10):
    ctx.reduce_first_order(f, n, k, 1, acc)
r1 = [ms for nm, ms in ctx.profile_end() if nm == "reduce_first"]
for _ in range(3):
    ctx.reduce_first_second_order(f, n, k, 1, acc, gram)
ctx.sync()
ctx.profile_begin(256)
for _ in range(10):
    ctx.reduce_first_second_order(f, n, k, 1, acc, gram)
recs = ctx.profile_end()
rb = [ms for nm, ms in recs if nm == "second_order"]
whole = sum(ms for nm, ms in recs) / 10.0     # every kernel of the call: Gram / first-order kernels, folds
print("first order alone %.3f ms; both orders in one call: %.3f ms of kernels (Gram kernel %.3f, min %.3f)" %
      (np.median(r1), whole, np.median(rb), min(rb)))
print("second order k=%d n=2^%d: %.3f ms (min %.3f)" % (k, int(np.log2(n)), np.median(r), min(r)))
