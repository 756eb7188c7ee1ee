"""hipMalloc / first-touch / hipFree latency by buffer size (why streamed passes are capped at 8 GiB)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
ctx = _native.Context(0)
for gb in (1, 4, 8, 16, 32, 64):
    for rep in range(2):
        t0 = time.time(); b = ctx.alloc(gb << 30); ctx.sync(); t1 = time.time()
        ctx.memset(b, 0); ctx.sync(); t2 = time.time()
        b.free(); ctx.sync(); t3 = time.time()
        print("%3d GiB: alloc %.1f ms, first memset %.1f ms, free %.1f ms" % (gb, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)), flush=True)
