"""Leak check of the host-boundary calls (round 5): free HBM before and after 300 rounds of upload_values / download_values /
sample rows / a host-model analysis / a loaded-values analysis with NaN rows."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varsens_amd as vs
from varsens_amd.saltelli import default_context
ctx = default_context()
def free(): return torch.cuda.mem_get_info()[0]
k, n, l = 5, 3000, 2
vals = np.random.default_rng(0).standard_normal((2 * n * (k + 1), l)); vals[7, 0] = np.nan
def once(i):
    o = vs.Objective(k, n + (i % 3), objective_vals=np.resize(vals, (2 * (n + (i % 3)) * (k + 1), l)), verbose=False)
    v = vs.Varsens(o, verbose=False)
    o.flat()
    s = vs.Sample(k, 500, vs.scale.Identity(), verbose=False)
    w = vs.Varsens(vs.Objective(k, 500, s, lambda X: X.sum(axis=1), verbose=False, vectorized=True), verbose=False)
    s.flat()
once(0); ctx.sync()
f0 = free(); t0 = time.time()
for i in range(300):
    once(i)
ctx.sync()
print("300 rounds in %.1f s; free HBM before %.1f MB, after %.1f MB, delta %.2f MB" % (time.time() - t0, f0 / 1e6, free() / 1e6, (f0 - free()) / 1e6))
