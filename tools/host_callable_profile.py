"""Where the wall time of the host-callable route goes (tools/host_routes.py: host_callable_route): cProfile of
Sample -> Objective(vectorised numpy model) -> Varsens at k = 32, n = 2^LOG2N, twice (first call pays the pinned ring)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import varsens_amd as vs
from varsens_amd import _native
ctx = _native.Context(0)
k, n = 32, 1 << int(os.environ.get("LOG2N", 18))
c = np.arange(1, k + 1, dtype=float)
def model(X):
    return X @ c
def run():
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False, context=ctx)
    o = vs.Objective(k, n, s, model, verbose=False, vectorized=True)
    return vs.Varsens(o, verbose=False)
for rep in range(2):
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
    print("rep %d: %.1f ms" % (rep, 1e3 * (time.perf_counter() - t0)))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
