"""Launch-shape experiments for the scan kernels: each row is one context created under the given knobs.

    python tools/scan_sweep.py
"""
import os, sys
os.environ["VS_TUNING"] = "1"   # the engine honours its VS_* measurement knobs only under this switch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation


def ctx_with(env):
    for a, b in env.items():
        os.environ[a] = str(b)
    c = _native.Context(0)
    for a in env:
        del os.environ[a]
    return c


def timed(ctx, fn, reps=7):
    fn(); fn(); ctx.sync()
    ctx.profile_begin(4 * reps + 8)
    for _ in range(reps):
        fn()
    rec = ctx.profile_end()
    per = {}
    for name, ms in rec:
        per.setdefault(name, []).append(ms)
    return {a: float(np.median(b)) for a, b in per.items()}


CASES = [
    # (k, log2 n, what, env)
    (32, 22, "eval", {"VS_EVAL_MODE": 2}),
    (32, 22, "eval", {"VS_EVAL_MODE": 1}),
    (32, 22, "fused", {"VS_FUSED_WGS_PER_CU": 2}),
    (32, 22, "fused", {"VS_FUSED_WGS_PER_CU": 3}),
    (32, 22, "fused", {"VS_FUSED_WGS_PER_CU": 4}),
    (32, 22, "fused", {"VS_FUSED_WGS_PER_CU": 6}),
    (6, 24, "fused", {"VS_FUSED_WGS_PER_CU": 2}),
    (6, 24, "fused", {"VS_FUSED_WGS_PER_CU": 4}),
    (6, 24, "fused", {"VS_FUSED_WGS_PER_CU": 7}),
    (64, 21, "fused", {"VS_FUSED_WGS_PER_CU": 1}),
    (64, 21, "fused", {"VS_FUSED_WGS_PER_CU": 2}),
    (128, 20, "eval", {"VS_EVAL_MODE": 2}),
    (128, 20, "eval", {"VS_EVAL_MODE": 2, "VS_SCAN_SHAPE": 816}),
    (128, 20, "fused", {"VS_FUSED_WGS_PER_CU": 1}),
    (128, 20, "fused", {"VS_FUSED_VARIANT": 1, "VS_FUSED_WGS_PER_CU": 1}),
    (200, 19, "eval", {"VS_EVAL_MODE": 2}),
    (256, 19, "eval", {"VS_EVAL_MODE": 2}),
    (48, 21, "eval", {"VS_EVAL_MODE": 2}),
    (48, 21, "eval", {"VS_EVAL_MODE": 2, "VS_SCAN_SHAPE": 808}),
    (96, 20, "eval", {"VS_EVAL_MODE": 2}),
    (96, 20, "eval", {"VS_EVAL_MODE": 2, "VS_SCAN_SHAPE": 1608}),
]
for k, log2n, what, env in CASES:
    n = 1 << log2n
    ctx = ctx_with(env)
    perm = seed1_permutation(n)
    d = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm)
    o = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.linspace(0.5, 2.0, k))
    N = 2 * (k + 1) * n
    if what == "eval":
        f = ctx.alloc(8 * N)
        t = timed(ctx, lambda: ctx.eval(d, o, 0, n, f)).get("eval")
        f.free()
    else:
        acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
        t = timed(ctx, lambda: ctx.eval_reduce_first_order(d, o, 0, n, acc)).get("fused")
    print("k=%3d n=2^%d %-5s %-55s %7.3f ms  %7.1f G evals/s" % (k, log2n, what, env, t, N / t / 1e6), flush=True)
    d.free(); o.free(); ctx.close()
