"""Evaluation-kernel time by generator and objective (k=32, n=2^22 unless K/LOG2N are set)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation, sobol_directions
k, n = int(os.environ.get("K", 32)), 1 << int(os.environ.get("LOG2N", 22))
ctx = _native.Context(0)
perm = seed1_permutation(n)
f = ctx.alloc(8 * 2 * (k + 1) * n)
designs = {
    "halton": ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm),
    "halton+linear scale": ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, _native.SCALE_LINEAR,
                                             -np.ones(k), 2 * np.ones(k), perm),
    "sobol": ctx.design_create(_native.GEN_SOBOL, k, n, 4096, 0, None, None, perm, sobol_directions=sobol_directions(k)),
}
objs = {
    "linear": ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float)),
    "g-function": ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, np.linspace(0, 9, k)),
}
for dn, d in designs.items():
    for on, o in objs.items():
        for _ in range(3):
            ctx.eval(d, o, 0, n, f)
        ctx.sync()
        ctx.profile_begin(64)
        for _ in range(10):
            ctx.eval(d, o, 0, n, f)
        r = [ms for nm, ms in ctx.profile_end() if nm == "eval"]
        print("%-22s %-11s eval %.3f ms (%.0f G evals/s)" % (dn, on, np.median(r), 2 * (k + 1) * n / np.median(r) / 1e6))
