"""BASELINE configs[2]: Ishigami, k=3, n=2^22, linear scaling to [-pi, pi]: kernel times."""
import os, sys
os.environ["VS_TUNING"] = "1"   # the engine honours its VS_* measurement knobs only under this switch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
for env in ({}, {"VS_EVAL_SCRATCH": "1"}):
    os.environ.update(env)
    ctx = _native.Context(0)
    for a in env: del os.environ[a]
    d = ctx.design_create(_native.GEN_HALTON, 3, n, 61, 1, np.full(3, -np.pi), np.full(3, np.pi), seed1_permutation(n))
    o = ctx.objective_create(_native.OBJ_ISHIGAMI, 3, 1, [7.0, 0.1])
    f = ctx.alloc(8 * 8 * n); acc = ctx.alloc(8 * _native.acc_len(3, 1, 1)); gram = ctx.alloc(8 * 27)
    def step():
        ctx.eval(d, o, 0, n, f); ctx.reduce_first_second_order(f, n, 3, 1, acc, gram); ctx.eval_reduce_first_order(d, o, 0, n, acc)
    for _ in range(20): step()
    ctx.sync(); ctx.profile_begin(512)
    for _ in range(20): step()
    per = {}
    for name, ms in ctx.profile_end(): per.setdefault(name, []).append(ms)
    # per kernel name: ms per step (a name may occur more than once in a step: the fused call evaluates one row
    # for its variance shift -- rounds 1-2 took a median over that mixed list and reported half the evaluation time)
    print(env or "default", {a: round(float(np.sum(b)) / 20.0, 4) for a, b in per.items()}, flush=True)
    ctx.close()
# the default call (both orders) in one kernel: vs_eval_reduce_first_second_order
ctx = _native.Context(0)
d = ctx.design_create(_native.GEN_HALTON, 3, n, 61, 1, np.full(3, -np.pi), np.full(3, np.pi), seed1_permutation(n))
o = ctx.objective_create(_native.OBJ_ISHIGAMI, 3, 1, [7.0, 0.1])
f = ctx.alloc(8 * 8 * n); acc = ctx.alloc(8 * _native.acc_len(3, 1, 1)); gram = ctx.alloc(8 * _native.gram_len(3))
for name, fn in (("two kernels (vs_eval + vs_reduce_first_second_order)", lambda: (ctx.eval(d, o, 0, n, f), ctx.reduce_first_second_order(f, n, 3, 1, acc, gram))),
                 ("fused both orders", lambda: ctx.eval_reduce_first_second_order(d, o, 0, n, acc, gram))):
    for _ in range(10): fn()
    ctx.sync(); ctx.profile_begin(512)
    for _ in range(20): fn()
    recs = ctx.profile_end()
    per = {}
    for nm, ms in recs: per.setdefault(nm, []).append(ms)
    print("%-56s %.3f ms of kernels per call" % (name, sum(ms for _, ms in recs) / 20.0),
          {a: round(float(np.median(b)) * len(b) / 20.0, 4) for a, b in per.items()}, flush=True)
