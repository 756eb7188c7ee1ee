import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import varsens_amd as vs
from varsens_amd.saltelli import default_context
ctx = default_context()
k, n = 32, 1 << 18
s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
buf = ctx.alloc(8 * n * k)
for blk, name in ((0, "M_1"), (1, "M_2"), (5, "N_j[3]"), (40, "N_nj[6]")):
    ctx.sample_flat(s._design, blk * n, n, buf); ctx.sync()
    ctx.profile_begin(16)
    for _ in range(3):
        ctx.sample_flat(s._design, blk * n, n, buf)
    r = [ms for nm, ms in ctx.profile_end()]
    print("%-8s sample_flat of 2^18 rows x 32: %.3f ms (%.1f GB/s)" % (name, np.median(r), 8 * n * k / np.median(r) / 1e6))
t0 = time.perf_counter(); x = ctx.sample_rows(s._design, 0, n, k); t1 = time.perf_counter()
print("sample_rows (64 MB to host): %.2f ms = %.1f GB/s" % (1e3 * (t1 - t0), 8 * n * k / (t1 - t0) / 1e9))
t0 = time.perf_counter(); x = ctx.sample_rows(s._design, 2 * n, 8 * n, k); t1 = time.perf_counter()
print("sample_rows (512 MB to host): %.2f ms = %.1f GB/s" % (1e3 * (t1 - t0), 8 * 8 * n * k / (t1 - t0) / 1e9))
