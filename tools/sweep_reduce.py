"""Sweep the reduction kernel's launch shape (16-byte loads per thread x workgroups per CU).
The knobs are read when a context is created, so every setting gets its own context."""
import os, sys
os.environ["VS_TUNING"] = "1"   # the engine honours its VS_* measurement knobs only under this switch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k = int(os.environ.get("K", 32)); n = 1 << int(os.environ.get("LOG2N", 22))
N = 2 * (k + 1) * n
perm = seed1_permutation(n)
for nv in (2, 4, 8):
    for wpc in (1, 2, 3, 4, 6, 8):
        os.environ["VS_REDUCE_NV"] = str(nv); os.environ["VS_REDUCE_WGS_PER_CU"] = str(wpc)
        ctx = _native.Context(0)
        design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm)
        obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
        f = ctx.alloc(8 * N); acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
        ctx.eval(design, obj, 0, n, f)
        for _ in range(2): ctx.reduce_first_order(f, n, k, 1, acc)
        ctx.sync(); ctx.profile_begin(256)
        for _ in range(12): ctx.reduce_first_order(f, n, k, 1, acc)
        r = ctx.profile_end()
        red = [ms for name, ms in r if name == "reduce_first"]; fold = [ms for name, ms in r if name == "fold"]
        print("nv=%d wgs/cu=%d: reduce %.4f ms (min %.4f) = %.0f GB/s ; fold %.4f" %
              (nv, wpc, np.median(red), min(red), 8 * N / np.median(red) / 1e6, np.median(fold)), flush=True)
        f.free(); acc.free(); design.free(); obj.free(); ctx.close()
