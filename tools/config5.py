"""BASELINE configs[4] on ONE MI355X: polynomial objective, k=128, n=2^24 (34.6 GB of f-values)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation

k, n = 128, 1 << int(os.environ.get("LOG2N", 24))
beta = 1.0 / (1.0 + np.arange(k) / 8.0)
ctx = _native.Context(0)
t0 = time.time(); perm = seed1_permutation(n); print("perm %.2fs" % (time.time() - t0), flush=True)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm)
obj = ctx.objective_create(_native.OBJ_POLYNOMIAL, k, 1, beta)
N = 2 * (k + 1) * n
f = ctx.alloc(8 * N)
acc = ctx.alloc(8 * _native.acc_len(k, 1, 1)); out = ctx.alloc(8 * _native.first_order_len(k, 1))
gram = ctx.alloc(8 * _native.gram_len(k))
ctx.profile_begin(256)
for _ in range(5):
    ctx.eval(design, obj, 0, n, f)
    ctx.reduce_first_order(f, n, k, 1, acc)
    ctx.finish_first_order(acc, n, k, 1, out)
    ctx.reduce_second_order(f, n, k, 1, gram)
    ctx.reduce_first_second_order(f, n, k, 1, acc, gram)
rec = ctx.profile_end()
# medians over five rounds (the first is cold: clocks, page tables); "both_orders" = every kernel of the one-call form
per = {}
for name, ms in rec:
    per.setdefault(name, []).append(ms)
calls = 5
last = {"eval": float(np.median(per["eval"])), "reduce_first": float(np.median(per["reduce_first"][0::2])),
        "second_order": float(np.median(per["second_order"][0::2])),
        "both_orders_one_call_gram_kernel": float(np.median(per["second_order"][1::2])),
        "fold": float(np.median(per["fold"])), "finish": float(np.median(per["finish"])),
        "second_fold": float(np.median(per["second_fold"]))}
print({a: round(b, 3) for a, b in last.items()}, flush=True)
nbl = (k + 15) // 16   # S S^T and T T^T, upper block triangles: nbl (nbl + 1) tiles of 16 x 16 per 4 sample rows
print("eval %.1f G evals/s ; reduce %.0f GB/s (%.1f%% of 8 TB/s) ; second order %.0f GB/s algorithmic, %.1f TFLOP/s (%d MFMA tiles of S S^T, T T^T)" %
      (N / last["eval"] / 1e6, 8 * N / last["reduce_first"] / 1e6, 100 * 8 * N / last["reduce_first"] / 1e6 / 8000,
       16.0 * k * n / last["second_order"] / 1e6, nbl * (nbl + 1) * (n / 4.0) * 2048 / last["second_order"] / 1e9, nbl * (nbl + 1)))
fin = ctx.download(out, (2 + 4 * k, 1))
V = beta ** 2 / 12.0
total = np.prod(1.0 + V) - 1.0           # closed form for a product of independent factors
S = V / total
ST = V * (np.prod(1.0 + V) / (1.0 + V)) / total
print("var_y %.6e (analytic %.6e)  max|sens-S| %.2e  max|sens_t-ST| %.2e" %
      (fin[1, 0], total, np.max(np.abs(fin[2 + 2 * k:2 + 3 * k, 0] - S)), np.max(np.abs(fin[2 + 3 * k:, 0] - ST))))
# the default call -- evaluation and both orders -- in ONE kernel, no f-values in HBM (vs_eval_reduce_first_second_order)
f.free()
if ctx.eval_reduce_both_supported(design, obj):
    g2 = ctx.alloc(8 * _native.gram_len(k)); acc2 = ctx.alloc(8 * _native.acc_len(k, 1, 1))
    for _ in range(2):
        ctx.eval_reduce_first_second_order(design, obj, 0, n, acc2, g2)
    ctx.sync()
    ctx.profile_begin(64)
    for _ in range(5):
        ctx.eval_reduce_first_second_order(design, obj, 0, n, acc2, g2)
    recs = ctx.profile_end()
    fused = [ms for nm, ms in recs if nm == "fused"]
    G1, G2 = ctx.download(gram, (2, k, k)), ctx.download(g2, (2, k, k))
    print("fused evaluate + both orders: kernel %.3f ms (min %.3f), whole call %.3f ms of kernels; max rel. difference of "
          "the Gram sums vs the two-kernel path %.1e" % (np.median(fused), min(fused), sum(ms for _, ms in recs) / 5.0,
                                                         float(np.max(np.abs(G2 - G1)) / np.max(np.abs(G1)))), flush=True)
    print("   = %.1f TFLOP/s of fp64 MFMA beside the evaluation arithmetic (pipe shared with the vector fp64 instructions)" %
          (nbl * (nbl + 1) * (n / 4.0) * 2048 / np.median(fused) / 1e9))
