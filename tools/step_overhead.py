"""What the per-launch HIP events cost the bench step: ms per step with every kernel bracketed, with only the
sens/sens_t reduction bracketed (VS_PROFILE_MASK), and with none.  python tools/step_overhead.py"""
import os, sys, time
os.environ["VS_TUNING"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation
k, n = 32, 1 << 22
perm = seed1_permutation(n)
for label, mask, prof in (("all kernels bracketed", None, True), ("reduction only", str(1 << 5), True), ("no events", None, False)):
    if mask is None:
        os.environ.pop("VS_PROFILE_MASK", None)
    else:
        os.environ["VS_PROFILE_MASK"] = mask
    ctx = _native.Context(0)
    d = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, perm)
    o = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
    N = 2 * (k + 1) * n
    fb = [ctx.alloc(8 * N) for _ in range(3)]
    nacc = _native.acc_len(k, 1, 1)
    accs = [ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)]
    outs = [ctx.alloc(8 * _native.first_order_len(k, 1)) for _ in range(2)]
    cnt = [0]
    def step():
        i = cnt[0]; b = i & 1; cnt[0] += 1
        ctx.stream_select(0)
        ctx.eval(d, o, 0, n, fb[i % 3])
        ctx.stream_wait_event(0, b)
        ctx.reduce_first_order(fb[(i - 1) % 3], n, k, 1, accs[b], 0, 0, 1)
        ctx.event_record(2 + b, 0); ctx.stream_wait_event(1, 2 + b); ctx.stream_select(1)
        ctx.finish_first_order(accs[b], n, k, 1, outs[b], 1)
        ctx.event_record(b, 1); ctx.stream_select(0)
    ctx.eval(d, o, 0, n, fb[2])
    for _ in range(400): step()
    ctx.sync()
    steps = 300
    if prof: ctx.profile_begin(10 * steps)
    t0 = time.perf_counter()
    for _ in range(steps): step()
    ctx.sync()
    dt = time.perf_counter() - t0
    recs = ctx.profile_end() if prof else []
    per = {}
    for nm, ms in recs: per.setdefault(nm, []).append(ms)
    print("%-24s %.4f ms per step   %s" % (label, 1e3 * dt / steps, {a: round(float(np.mean(v)), 4) for a, v in per.items()}), flush=True)
    ctx.close()
