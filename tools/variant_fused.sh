#!/bin/bash
for v in "" $(ls build_variants/*.so); do
  echo "== variant: ${v:-default}"
  for w in 2 3 4; do
    VS_FUSED_ONLY=1 VARSENS_HIP_LIB=$v W=$w timeout -k 10 100 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = 32, 1 << 22
w = os.environ["W"]
os.environ["VS_FUSED_WGS_PER_CU"] = w
ctx = _native.Context(0)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
for _ in range(3):
    ctx.eval_reduce_first_order(design, obj, 0, n, acc)
ctx.sync()
ctx.profile_begin(64)
for _ in range(10):
    ctx.eval_reduce_first_order(design, obj, 0, n, acc)
r = [ms for name, ms in ctx.profile_end() if name == "fused"]
print("  wgs/cu=%s fused %.3f ms (min %.3f)" % (w, np.median(r), min(r)))
PY
  done
done
