"""Evaluation kernel fed from an uploaded (RAW) design, e.g. an opaque scaling lambda (tuning aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = 32, 1 << int(os.environ.get("LOG2N", 21))
ctx = _native.Context(0)
raw = np.random.RandomState(0).rand(2 * n, k)
t0 = time.time()
d = ctx.design_create(_native.GEN_RAW, k, n, 0, 0, None, None, seed1_permutation(n), raw)
print("design_create (upload %.1f GB): %.2f s" % (raw.nbytes / 1e9, time.time() - t0))
o = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
f = ctx.alloc(8 * 2 * (k + 1) * n)
ctx.eval(d, o, 0, n, f); ctx.sync(); ctx.profile_begin(16)
for _ in range(4): ctx.eval(d, o, 0, n, f)
ms = np.median([m for nm, m in ctx.profile_end()])
print("eval RAW k=%d n=2^%d: %.3f ms (%.1f G evals/s)" % (k, int(np.log2(n)), ms, 2 * (k + 1) * n / ms / 1e6))
