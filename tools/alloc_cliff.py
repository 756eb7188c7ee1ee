import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, cProfile, pstats
import varsens_amd as vs
from varsens_amd import _native
from varsens_amd.saltelli import default_context
ctx = default_context()
for gb in (1.0, 2.0, 3.0, 3.9, 4.1, 6.0):
    for rep in range(2):
        t0 = time.time(); b = ctx.alloc(int(gb * (1 << 30))); ctx.sync(); t1 = time.time()
        ctx.memset(b, 0); ctx.sync(); t2 = time.time()
        b.free(); ctx.sync(); t3 = time.time()
        print("%.1f GiB: alloc %.1f ms, first memset %.1f ms, free %.1f ms" % (gb, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)), flush=True)
k = 192
for n in (655360, 1310720):
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    a = np.linspace(0, 99, k)
    vs.Varsens(vs.objectives.GFunction(a), sample=s, verbose=False)
    pr = cProfile.Profile(); pr.enable()
    t0 = time.time()
    v = vs.Varsens(vs.objectives.GFunction(a), sample=s, verbose=False)
    print("n=%d: %.1f ms, stream_rows=%s" % (n, 1e3 * (time.time() - t0), v._stream_rows))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
