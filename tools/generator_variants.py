"""Evaluation / fused kernels by generator and scaling (the scan kernels' specialised vs general column code):
    python tools/generator_variants.py [K] [LOG2N]"""
import os, sys
os.environ["VS_TUNING"] = "1"   # the engine honours its VS_* measurement knobs only under this switch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation, sobol_directions
k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
N = 2 * (k + 1) * n
perm = seed1_permutation(n)
lo, hi = np.full(k, -1.0), np.full(k, 2.0)
plo, phi = np.full(k, 0.5), np.full(k, 8.0)


def run(label, env, make):
    for a, b in env.items():
        os.environ[a] = str(b)
    ctx = _native.Context(0)
    for a in env:
        del os.environ[a]
    d = make(ctx)
    o = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, np.linspace(0.0, 9.0, k))
    f = ctx.alloc(8 * N)
    acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
    out = []
    for what in ("eval", "fused"):
        if what == "fused" and not ctx.eval_reduce_supported(d, o):
            out.append("fused   -   ")
            continue
        fn = (lambda: ctx.eval(d, o, 0, n, f)) if what == "eval" else (lambda: ctx.eval_reduce_first_order(d, o, 0, n, acc))
        fn(); fn(); ctx.sync()
        ctx.profile_begin(64)
        for _ in range(6):
            fn()
        ms = np.median([t for name, t in ctx.profile_end() if name == what and t > 0.02])
        out.append("%s %7.3f ms" % (what, ms))
    print("%-44s %s" % (label, "   ".join(out)), flush=True)
    f.free(); ctx.close()


H = _native.GEN_HALTON
run("halton identity (specialised)", {}, lambda c: c.design_create(H, k, n, 1 + 20 * k, 0, None, None, perm))
run("halton identity (general column code)", {"VS_SCAN_GENERAL": 1}, lambda c: c.design_create(H, k, n, 1 + 20 * k, 0, None, None, perm))
run("halton linear (specialised)", {}, lambda c: c.design_create(H, k, n, 1 + 20 * k, 1, lo, hi, perm))
run("halton power (general)", {}, lambda c: c.design_create(H, k, n, 1 + 20 * k, 2, plo, phi, perm))
run("sobol identity (general)", {}, lambda c: c.design_create(_native.GEN_SOBOL, k, n, 4096, 0, None, None, perm, None, sobol_directions(k)))
run("halton identity, strict order (chain kernel)", {"VS_EVAL_MODE": 1}, lambda c: c.design_create(H, k, n, 1 + 20 * k, 0, None, None, perm))
