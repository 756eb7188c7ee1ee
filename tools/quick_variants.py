"""Time the evaluation kernel for other generators / scalings / objectives (tuning aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation, sobol_directions

k, n = 32, 1 << 22
ctx = _native.Context(0)
perm = seed1_permutation(n)
N = 2 * (k + 1) * n
f = ctx.alloc(8 * N)
lo, hi = np.full(k, 1.0), np.full(k, 10.0)


def timed(fn, reps=5):
    fn(); ctx.sync()
    ctx.profile_begin(64)
    for _ in range(reps):
        fn()
    return np.median([ms for name, ms in ctx.profile_end() if name == "eval"])


designs = {
    "halton/identity": ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm),
    "halton/linear": ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 1, lo, hi, perm),
    "halton/power": ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 2, lo, hi, perm),
    "sobol/identity": ctx.design_create(_native.GEN_SOBOL, k, n, 4096, 0, None, None, perm, None, sobol_directions(k)),
}
objs = {
    "linear": ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float)),
    "gfunc": ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, np.linspace(0, 9, k)),
    "expr-linear": ctx.objective_create_source(k, 1, "double s=0.0; for(int i=0;i<k;++i) s = s + p[i]*x[i]; out[0]=s;",
                                               np.arange(1, k + 1, dtype=float)),
}
for dn, d in designs.items():
    for on, o in objs.items():
        if dn != "halton/identity" and on != "linear":
            continue
        ms = timed(lambda: ctx.eval(d, o, 0, n, f))
        print("%-16s %-12s %.3f ms  %.1f G evals/s" % (dn, on, ms, N / ms / 1e6), flush=True)
