"""Does pass i's reduction (HBM-read-bound) run beside pass i+1's evaluation (vector-issue-bound)?
Two f-value buffers, two streams; compares the serial schedule with the overlapped one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = int(os.environ.get("K", 32)), 1 << int(os.environ.get("LOG2N", 22))
steps = int(os.environ.get("STEPS", 200))
ctx = _native.Context(0)
design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
N = 2 * (k + 1) * n
f = [ctx.alloc(8 * N), ctx.alloc(8 * N)]
nacc = _native.acc_len(k, 1, 1)
acc = [ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)]
out = [ctx.alloc(8 * _native.first_order_len(k, 1)) for _ in range(2)]


def serial(i):
    ctx.stream_select(0)
    ctx.eval(design, obj, 0, n, f[0])
    ctx.reduce_first_order(f[0], n, k, 1, acc[0])
    ctx.finish_first_order(acc[0], n, k, 1, out[0])


def overlapped(i):
    b = i & 1
    ctx.stream_select(0)
    ctx.stream_wait_event(0, b)          # the reduction of pass i-2 is done with f[b]
    ctx.eval(design, obj, 0, n, f[b])
    ctx.event_record(2 + b, 0)
    ctx.stream_select(1)
    ctx.stream_wait_event(1, 2 + b)
    ctx.reduce_first_order(f[b], n, k, 1, acc[b])
    ctx.finish_first_order(acc[b], n, k, 1, out[b])
    ctx.event_record(b, 1)
    ctx.stream_select(0)


for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    for i in range(20):
        fn(i)
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    r = ctx.download(out[(steps - 1) & 1] if fn is overlapped else out[0], (2 + 4 * k, 1))
    print("%-10s %.3f ms per pass = %.1f GSamples/s   sens[0]=%.15g" % (name, dt * 1e3, N / dt / 1e9, r[2 + 2 * k, 0]), flush=True)
