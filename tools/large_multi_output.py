"""A shape like the reference's model studies: k=100 parameters, l=10 outputs, n=20000 draws, a user
objective compiled by hiprtc -- first and second order (k*l = 1000 index rows, 528 super-block pairs) against
the oracle's numpy arithmetic on the same f-values."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import varsens_amd as vs
from oracle import saltelli_oracle as orc   # tools/ is measurement tooling, not the product

k, l, n = int(os.environ.get("K", 100)), int(os.environ.get("L", 10)), int(os.environ.get("N", 20000))
body = """
for (int o = 0; o < %d; ++o) {
    double s = 0.0;
    for (int c = 0; c < k; ++c) s += p[(o * 7 + c) %% k] * x[c] * (1.0 + 0.01 * o * x[(c + o + 1) %% k]);
    out[o] = s;
}""" % l
params = 1.0 / (1.0 + np.arange(k))
obj = vs.objectives.Expression(body, outputs=l, params=params)
t0 = time.time()
s = vs.Sample(k, n, vs.scale.Linear(-1.0, 2.0), verbose=False)
o = vs.Objective(k, n, s, obj, verbose=False)
t1 = time.time()
v = vs.Varsens(o, verbose=False)
t2 = time.time()
v = vs.Varsens(o, verbose=False)
t3 = time.time()
print("k=%d l=%d n=%d: design + hiprtc + evaluation of %.2e values %.2f s; reductions + finish %.3f s (second call %.3f s)"
      % (k, l, n, 2.0 * (k + 1) * n * l, t1 - t0, t2 - t1, t3 - t2), flush=True)
t0 = time.time()
R = orc.compute_varsens(o.fM_1, o.fM_2, o.fN_j, o.fN_nj, n)
print("oracle (numpy, default BLAS threads): %.1f s" % (time.time() - t0), flush=True)
for name in ("E_2", "var_y", "U_j", "U_nj", "sens", "sens_t", "sens_2", "sens_2n"):
    a, b = getattr(v, name), R[name]
    scale = np.abs(b).max()
    print("%-7s shape %-18s max |diff| / max |ref| = %.2e" % (name, a.shape, np.abs(a - b).max() / scale))
