"""One shape of the evaluation / fused kernels, a few launches (for rocprofv3 PMC passes):
    python tools/eval_one.py K LOG2N [eval|fused] [kind] [halton|sobol]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation, sobol_directions
k, log2n = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else "eval"
kind = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[sys.argv[4] if len(sys.argv) > 4 else "poly"]
n = 1 << log2n
ctx = _native.Context(0)
if len(sys.argv) > 5 and sys.argv[5] == "sobol":
    d = ctx.design_create(_native.GEN_SOBOL, k, n, 4096, 0, None, None, seed1_permutation(n), None, sobol_directions(k))
else:
    d = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
o = ctx.objective_create(kind, k, 1, np.linspace(0.5, 2.0, k))
N = 2 * (k + 1) * n
f = ctx.alloc(8 * N) if what == "eval" else None
acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
ctx.profile_begin(64)
for _ in range(4):
    if what == "eval":
        ctx.eval(d, o, 0, n, f)
    else:
        ctx.eval_reduce_first_order(d, o, 0, n, acc)
rec = [ms for name, ms in ctx.profile_end() if name in ("eval", "fused") and ms > 0.02]
print("k=%d n=2^%d %s: %.3f ms  %.1f G evals/s" % (k, log2n, what, np.median(rec), N / np.median(rec) / 1e6))
