"""A design whose f-values do not fit the GPU: polynomial objective, k=128, n=2^27 (277 GB of
f-values; 288 GB of HBM) through Varsens' streamed row ranges (rows_per_pass='auto')."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import varsens_amd as vs

k, n = int(os.environ.get("K", 128)), 1 << int(os.environ.get("LOG2N", 27))
beta = 1.0 / (1.0 + np.arange(k) / 8.0)
t0 = time.time()
sample = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
t1 = time.time()
print("design (host seed-1 permutation of %d rows, upload): %.1f s" % (n, t1 - t0), flush=True)
v = vs.Varsens(vs.objectives.Polynomial(beta), sample=sample, verbose=False, second_order=False)
t2 = time.time()
N = 2 * (k + 1) * n
print("k=%d n=2^%d: %.3e evaluations, %.1f GB of f-values, %d draws per pass; evaluate + reduce + finish %.2f s "
      "= %.1f GSamples/s" % (k, int(np.log2(n)), N, 8 * N / 1e9, v._stream_rows, t2 - t1, N / (t2 - t1) / 1e9), flush=True)
V = beta ** 2 / 12.0
total = np.prod(1.0 + V) - 1.0
S = V / total
ST = V * (np.prod(1.0 + V) / (1.0 + V)) / total
print("var_y %.6e (analytic %.6e)  max|sens-S| %.2e  max|sens_t-ST| %.2e  rows kept %d" %
      (v.var_y[0], total, np.max(np.abs(v.sens[:, 0] - S)), np.max(np.abs(v.sens_t[:, 0] - ST)), v.accumulators[0]))
# where the wall time goes: the same call again (allocator warm), and the kernels alone
for rep in range(4):
    t3 = time.time()
    v2 = vs.Varsens(vs.objectives.Polynomial(beta), sample=sample, verbose=False, second_order=False)
    t4 = time.time()
    print("call %d: %.2f s = %.1f GSamples/s" % (rep + 2, t4 - t3, N / (t4 - t3) / 1e9), flush=True)
ctx = sample._ctx
ctx.profile_begin(256)
v3 = vs.Varsens(vs.objectives.Polynomial(beta), sample=sample, verbose=False, second_order=False)
rec = ctx.profile_end()
tot = {}
for name, ms in rec:
    tot[name] = tot.get(name, 0.0) + ms
print("kernel time by name (ms):", {a: round(b, 1) for a, b in tot.items()}, "sum %.1f ms" % sum(tot.values()))
