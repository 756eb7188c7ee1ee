"""Evaluation kernel variants side by side: chain scheme (VS_EVAL_MODE=1) vs segmented scan (=2), and the
fused evaluate+reduce kernels, for several k.  One context per mode (the knobs are read at context creation).

    python tools/eval_modes.py [log2 of total f-values, default 28]
"""
import os, sys, time
os.environ["VS_TUNING"] = "1"   # the engine honours its VS_* measurement knobs only under this switch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation

budget = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
KS = [int(x) for x in os.environ.get("KS", "6,16,32,48,64,96,128,200,256").split(",")]
KIND = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[os.environ.get("KIND", "lin")]


def ctx_with(**env):
    for a, b in env.items():
        os.environ[a] = str(b)
    c = _native.Context(0)
    for a in env:
        del os.environ[a]
    return c


def timed(ctx, fn, reps=5):
    fn(); ctx.sync()
    ctx.profile_begin(4 * reps + 8)
    for _ in range(reps):
        fn()
    rec = ctx.profile_end()
    per = {}
    for name, ms in rec:
        per.setdefault(name, []).append(ms)
    return {a: float(np.median(b)) for a, b in per.items()}


modes = {"chain": ctx_with(VS_EVAL_MODE=1), "scan": ctx_with(VS_EVAL_MODE=2)}
if os.environ.get("FUSED_VARIANT"):
    modes["scan_v1"] = ctx_with(VS_EVAL_MODE=2, VS_FUSED_VARIANT=1)
print("%5s %9s | %-28s | %-28s | %s" % ("k", "n", "eval ms chain / scan", "fused ms chain / scan", "G evals/s scan eval, fused"), flush=True)
for k in KS:
    n = max(1 << 10, 1 << int(np.floor(np.log2(budget / (2 * (k + 1))))))
    perm = seed1_permutation(n)
    params = np.linspace(0.5, 2.0, k)
    N = 2 * (k + 1) * n
    out = {}
    for name, ctx in modes.items():
        d = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm)
        o = ctx.objective_create(KIND, k, 1, params)
        f = ctx.alloc(8 * N)
        acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
        t = timed(ctx, lambda: ctx.eval(d, o, 0, n, f))
        out[name, "eval"] = t.get("eval")
        if ctx.eval_reduce_supported(d, o) and not (name == "chain" and k > 32):
            t = timed(ctx, lambda: ctx.eval_reduce_first_order(d, o, 0, n, acc))
            out[name, "fused"] = t.get("fused")
        f.free(); d.free(); o.free()
    fm = lambda v: "  -   " if v is None else "%6.3f" % v
    extra = ""
    if "scan_v1" in modes:
        extra = " | fused v1 %s" % fm(out.get(("scan_v1", "fused")))
    print("%5d %9d | %s / %s            | %s / %s            | %7.1f %7.1f%s" % (
        k, n, fm(out["chain", "eval"]), fm(out["scan", "eval"]), fm(out.get(("chain", "fused"))), fm(out.get(("scan", "fused"))),
        N / out["scan", "eval"] / 1e6, N / out["scan", "fused"] / 1e6 if out.get(("scan", "fused")) else 0.0, extra), flush=True)
