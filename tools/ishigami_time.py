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
    print(env or "default", {a: round(float(np.median([x for x in b if x > 0.005] or b)), 4) for a, b in per.items()}, flush=True)
    ctx.close()
