import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import varsens_amd as vs
a = [0, .5, 3, 9, 99, 99]
for k, n in ((6, 1024), (6, 51200), (32, 4096), (6, 1 << 20), (32, 1 << 22)):
    aa = a if k == 6 else np.linspace(0, 9, k)
    for rep in range(2):
        t0 = time.perf_counter()
        N = 50
        for _ in range(N):
            v = vs.Varsens(vs.objectives.GFunction(aa), vs.scale.Identity(), k, n, verbose=False)
        dt = (time.perf_counter() - t0) / N
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    t0 = time.perf_counter()
    for _ in range(N):
        v = vs.Varsens(vs.objectives.GFunction(aa), sample=s, verbose=False)
    dt2 = (time.perf_counter() - t0) / N
    print("k=%d n=%d: Varsens(objective, scaling, k, n) %.3f ms per call; with a prepared Sample %.3f ms" % (k, n, 1e3 * dt, 1e3 * dt2))
import cProfile, pstats
s = vs.Sample(6, 1024, vs.scale.Identity(), verbose=False)
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    v = vs.Varsens(vs.objectives.GFunction(a), sample=s, verbose=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# the same for a large call (BASELINE configs[3]'s shape on one GPU): where the wall time beyond the kernels goes
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    v = vs.Varsens(vs.objectives.Linear(np.arange(1, 33.0)), vs.scale.Identity(), 32, 1 << 22, verbose=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
