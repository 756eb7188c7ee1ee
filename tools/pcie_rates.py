"""Host<->device copy rates of vs_upload / vs_download from pageable numpy memory."""
import sys, time
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from varsens_amd import _native
ctx = _native.Context(0)
for mb in (64, 512, 2200):
    a = np.random.rand(mb * 131072)
    b = ctx.upload(a); ctx.sync()
    t0 = time.time(); ctx.upload_into(b, a); ctx.sync(); t1 = time.time()
    out = ctx.download(b, a.shape); t2 = time.time()
    print("%5d MB: upload %.1f GB/s, download %.1f GB/s" % (mb, a.nbytes / (t1 - t0) / 1e9, a.nbytes / (t2 - t1) / 1e9), flush=True)
    b.free()
