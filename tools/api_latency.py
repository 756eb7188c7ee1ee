"""Wall time of whole API calls (includes Python, allocation, kernels, downloads)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import varsens_amd as vs
a = [0, 0.5, 3, 9, 99, 99]
def g_objective(x):
    return np.prod((np.abs(4.0 * x - 2.0) + a) / (1.0 + np.array(a)))
for label, fn in (
    ("config1 device objective k=6 n=1024 (first+second order)", lambda: vs.Varsens(vs.objectives.GFunction(a), vs.scale.Identity(), 6, 1024, verbose=False)),
    ("config1 python callable    k=6 n=1024", lambda: vs.Varsens(g_objective, lambda x: x, 6, 1024, verbose=False)),
    ("config2 device objective k=6 n=2^20", lambda: vs.Varsens(vs.objectives.GFunction(a), vs.scale.Identity(), 6, 1 << 20, verbose=False)),
    ("config4 device objective k=32 n=2^22 first order (fused)", lambda: vs.Varsens(vs.objectives.Linear(np.arange(1, 33.0)), vs.scale.Identity(), 32, 1 << 22, verbose=False, second_order=False)),
    ("config4 device objective k=32 n=2^22 first+second order", lambda: vs.Varsens(vs.objectives.Linear(np.arange(1, 33.0)), vs.scale.Identity(), 32, 1 << 22, verbose=False)),
):
    fn()
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); v = fn(); t.append(time.perf_counter() - t0)
    print("%-62s %.1f ms" % (label, 1e3 * min(t)), flush=True)
