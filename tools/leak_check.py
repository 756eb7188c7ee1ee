"""Thousands of small analyses in one process: free HBM before and after (leak check)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varsens_amd as vs
model = [0, .5, 3, 9, 99, 99]
def free(): return torch.cuda.mem_get_info()[0]
v = vs.Varsens(vs.objectives.GFunction(model), vs.scale.Identity(), 6, 1024, verbose=False)
f0 = free()
t0 = time.time()
for i in range(3000):
    v = vs.Varsens(vs.objectives.GFunction(model), vs.scale.Identity(), 6, 1024 + (i % 7), verbose=False)
    if i % 500 == 0:
        s = vs.Sample(5, 333, vs.scale.Linear(-1, 2), verbose=False)
        o = vs.Objective(5, 333, s, lambda x: float(np.sum(x)), verbose=False)
        w = vs.Varsens(o, verbose=False)
        e = vs.Varsens(vs.objectives.Expression("out[0] = x[0] + p[0]*x[1];", params=[2.0]), vs.scale.Identity(), 2, 4096, verbose=False)
print("3000 analyses in %.1f s; free HBM before %.1f MB, after %.1f MB, delta %.2f MB" % (time.time() - t0, f0 / 1e6, free() / 1e6, (f0 - free()) / 1e6))
