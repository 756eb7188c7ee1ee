"""Where a wave of the evaluation scan kernel spends its life: shader-clock stamps at start, before the barrier
and at the end (the barrier opens when the block's last wave arrives), from a diagnostic build (VS_EXTRA_CFLAGS=-DVS_SCAN_STAMPS python -m varsens_amd.build --force).
    python tools/scan_stamps.py K LOG2N [kind]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, log2n = int(sys.argv[1]), int(sys.argv[2])
kind = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[sys.argv[3] if len(sys.argv) > 3 else "poly"]
n = 1 << log2n
W = 8 if k > 48 else 4
ctx = _native.Context(0)
d = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
o = ctx.objective_create(kind, k, 1, np.linspace(0.5, 2.0, k))
N = 2 * (k + 1) * n
nblk = n // 64
nst = ((nblk + 511) // 512) * W * 4
f = ctx.alloc(8 * (N + nst))
for _ in range(3):
    ctx.eval(d, o, 0, n, f)
ctx.sync()
st = ctx.download(f, (nst,), byte_offset=8 * N).view(np.int64).reshape(-1, W, 4)
gen = st[:, :, 1] - st[:, :, 0]
release = st[:, :, 1].max(axis=1, keepdims=True)     # the barrier opens when the last wave arrives
bar = release - st[:, :, 1]
out = st[:, :, 3] - release
print("k=%d n=2^%d: %d sampled blocks x %d waves" % (k, log2n, st.shape[0], W))
print("generation  mean %.0f  per wave index %s" % (gen.mean(), np.round(gen.mean(axis=0)).astype(int)))
print("barrier     mean %.0f  per wave index %s" % (bar.mean(), np.round(bar.mean(axis=0)).astype(int)))
print("output walk mean %.0f  per wave index %s" % (out.mean(), np.round(out.mean(axis=0)).astype(int)))
print("whole wave  mean %.0f cycles; slowest generation of a block %.0f, fastest %.0f" %
      ((st[:, :, 3] - st[:, :, 0]).mean(), gen.max(axis=1).mean(), gen.min(axis=1).mean()))
