import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
ctx = _native.Context(0)
for k in (6, 32, 64, 128):
    for n in ((1 << 21), (1 << 21) + 1):
        d = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
        o = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
        f = ctx.alloc(8 * 2 * (k + 1) * n)
        ctx.eval(d, o, 0, n, f)
        acc, gram = ctx.alloc(8 * _native.acc_len(k, 1, 1)), ctx.alloc(8 * _native.gram_len(k))
        res = {}
        for name, fn in (("first", lambda: ctx.reduce_first_order(f, n, k, 1, acc)), ("second", lambda: ctx.reduce_second_order(f, n, k, 1, gram)),
                         ("both", lambda: ctx.reduce_first_second_order(f, n, k, 1, acc, gram)), ("eval", lambda: ctx.eval(d, o, 0, n, f)),
                         ("fused_both", lambda: ctx.eval_reduce_first_second_order(d, o, 0, n, acc, gram))):
            for _ in range(3): fn()
            ctx.sync(); ctx.profile_begin(256)
            for _ in range(8): fn()
            res[name] = sum(ms for _, ms in ctx.profile_end()) / 8.0
        print("k=%3d n=%8d: " % (k, n) + "  ".join("%s %.3f" % kv for kv in res.items()), flush=True)
        for h in (f, acc, gram, d, o): h.free()
