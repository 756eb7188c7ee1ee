"""The evaluation kernel INSIDE a sustained evaluate -> reduce step (three rotating f-value buffers, as bench.py), by k:
    K="16 32 48 64 96 128" LOG2N=22 python3 tools/eval_in_step.py      (n is halved per doubling of k beyond 32)
Launched alone the kernel can rank two variants the other way round (profiles/r04/ab_halton_forms.txt)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
ctx = _native.Context(0)
for k in [int(x) for x in os.environ.get("K", "16 32 48 64 96 128").split()]:
    lg = int(os.environ.get("LOG2N", 22))
    n = 1 << (lg if k <= 32 else lg - 1 if k <= 64 else lg - 2)
    d = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
    kind = os.environ.get("KIND", "lin")
    code, params = {"lin": (_native.OBJ_LINEAR, np.arange(1, k + 1, dtype=float)),
                    "g": (_native.OBJ_G_FUNCTION, np.linspace(0, 9, k)),
                    "poly": (_native.OBJ_POLYNOMIAL, 1.0 / (1.0 + np.arange(k) / 8.0))}[kind]
    obj = ctx.objective_create(code, k, 1, params)
    f = [ctx.alloc(8 * 2 * (k + 1) * n) for _ in range(3)]
    acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
    ctx.eval(d, obj, 0, n, f[2])
    def step(i):
        ctx.eval(d, obj, 0, n, f[i % 3])
        ctx.reduce_first_order(f[(i - 1) % 3], n, k, 1, acc)
    for i in range(60):
        step(i)
    ctx.sync()
    ctx.profile_begin(1024)
    for i in range(100):
        step(i)
    by = {}
    for nm, ms in ctx.profile_end():
        by.setdefault(nm, []).append(ms)
    print("k=%3d n=2^%d %s: eval %.4f ms (median %.4f)  reduce_first %.4f  -> %.1f G evals/s" % (
        k, int(np.log2(n)), kind, np.mean(by["eval"]), np.median(by["eval"]), np.mean(by["reduce_first"]),
        2 * (k + 1) * n / np.mean(by["eval"]) / 1e6), flush=True)
    for b in f + [acc, obj, d]:
        b.free()
