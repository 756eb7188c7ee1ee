"""Fused evaluate + both orders (vs_eval_reduce_first_second_order) beside vs_eval + vs_reduce_first_second_order:
median kernel times per call.   K=32 LOG2N=22 KIND=lin [L=4] [MODE=2] python3 tools/fused_both.py   (L: outputs; MODE: 0 identity, 1 linear, 2 power)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = int(os.environ.get("K", 32)), 1 << int(os.environ.get("LOG2N", 22))
kind = os.environ.get("KIND", "lin")
l = int(os.environ.get("L", 1))
ctx = _native.Context(0)
mode = int(os.environ.get("MODE", 0))      # 0 identity, 1 linear, 2 power scaling
lo, hi = (None, None) if mode == 0 else (np.full(k, 0.5), np.linspace(1.0, 3.0, k))
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, mode, lo, hi, seed1_permutation(n))
code, params = {"lin": (_native.OBJ_LINEAR, np.arange(1, k + 1, dtype=float)),
                "g": (_native.OBJ_G_FUNCTION, np.linspace(0, 9, k)),
                "poly": (_native.OBJ_POLYNOMIAL, 1.0 / (1.0 + np.arange(k) / 8.0))}[kind]
params = np.stack([params * (1.0 + 0.25 * o) for o in range(l)])
obj = ctx.objective_create(code, k, l, params)
f = ctx.alloc(8 * 2 * (k + 1) * n * l)
acc, gram = ctx.alloc(8 * _native.acc_len(k, l, 1)), ctx.alloc(8 * _native.gram_len(k * l))
def two():
    ctx.eval(design, obj, 0, n, f)
    ctx.reduce_first_second_order(f, n, k, l, acc, gram)
def one():
    ctx.eval_reduce_first_second_order(design, obj, 0, n, acc, gram)
for name, fn in (("two-kernel path", two), ("fused both orders", one)):
    for _ in range(5):
        fn()
    ctx.sync()
    ctx.profile_begin(512)
    for _ in range(20):
        fn()
    recs = ctx.profile_end()
    by = {}
    for nm, ms in recs:
        by.setdefault(nm, []).append(ms)
    total = sum(ms for _, ms in recs) / 20.0
    print("%-18s k=%d l=%d n=2^%d %s: %.3f ms of kernels per call  (%s)" % (
        name, k, l, int(np.log2(n)), kind, total,
        ", ".join("%s %.3f" % (nm, np.median(v) * len(v) / 20.0) for nm, v in by.items())), flush=True)
    g = ctx.download(gram, (2, k * l, k * l))
    print("   gram checksum %.15g  acc[7] %.15g" % (float(g.sum()), float(ctx.download(acc, (_native.acc_len(k, l, 1),))[7])))
