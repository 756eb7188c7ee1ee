"""Second-order sums of several outputs (K = k*l index rows per half): median kernel time per shape.

    SHAPES=16x4,32x4,20x3 LOG2N=22 python3 tools/gram_multi.py      (VARSENS_HIP_LIB=... for an A/B of two builds)
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
n = 1 << int(os.environ.get("LOG2N", 22))
ctx = _native.Context(0)
for shape in os.environ.get("SHAPES", "16x4,32x2,32x4,20x3,24x5").split(","):
    k, l = (int(v) for v in shape.split("x"))
    design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
    obj = ctx.objective_create(_native.OBJ_LINEAR, k, l, np.arange(1, k * l + 1, dtype=float))
    f = ctx.alloc(8 * 2 * (k + 1) * n * l)
    ctx.eval(design, obj, 0, n, f)
    gram = ctx.alloc(8 * _native.gram_len(k * l))
    for _ in range(3):
        ctx.reduce_second_order(f, n, k, l, gram)
    ctx.sync()
    ctx.profile_begin(64)
    for _ in range(10):
        ctx.reduce_second_order(f, n, k, l, gram)
    r = [ms for nm, ms in ctx.profile_end() if nm == "second_order"]
    g = ctx.download(gram, (2, k * l, k * l))
    print("k=%d l=%d (K=%d) n=2^%d: second order %.3f ms (min %.3f)  checksum %.17g" %
          (k, l, k * l, int(np.log2(n)), np.median(r), min(r), float(g.sum())), flush=True)
    for h in (gram, f):
        h.free()
