"""Run only the second-order Gram kernel a few times (for rocprofv3 counter passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = 32, 1 << 22
ctx = _native.Context(0)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
f = ctx.alloc(8 * 2 * (k + 1) * n)
ctx.eval(design, obj, 0, n, f)
gram = ctx.alloc(8 * 3 * k * k)
for _ in range(4):
    ctx.reduce_second_order(f, n, k, 1, gram)
ctx.sync()
