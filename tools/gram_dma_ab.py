"""Second order alone at 33 .. 64 index rows: gram_dma_kernel against the kernel it replaced (VS_GRAM_DMA=0), same f-values,
alternating rounds.  K=64 LOG2N=22 python tools/gram_dma_ab.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varsens_amd import _native
from varsens_amd.saltelli import seed1_permutation

ks = [int(x) for x in os.environ.get("K", "64").split(",")]
n = 1 << int(os.environ.get("LOG2N", 22))
os.environ["VS_TUNING"] = "1"
ctxs = {}
for name, v in (("gram_dma_kernel", "2"), ("gram_st_kernel", "0")):   # 2: also 17 .. 32 index rows (gram_diag_wide_kernel otherwise)
    os.environ["VS_GRAM_DMA"] = v
    ctxs[name] = _native.Context(0)
for k in ks:
    res = {}
    both = {}
    grams = {}
    for rnd in range(3):
        for name, ctx in ctxs.items():
            design = ctx.design_create(_native.GEN_HALTON, k, n, 1 + 20 * k, 0, None, None, seed1_permutation(n))
            obj = ctx.objective_create(_native.OBJ_LINEAR, k, 1, np.arange(1, k + 1, dtype=float))
            f = ctx.alloc(8 * 2 * (k + 1) * n)
            ctx.eval(design, obj, 0, n, f)
            gram = ctx.alloc(8 * _native.gram_len(k))
            for _ in range(3):
                ctx.reduce_second_order(f, n, k, 1, gram)
            ctx.sync()
            ctx.profile_begin(64)
            for _ in range(10):
                ctx.reduce_second_order(f, n, k, 1, gram)
            r = [ms for nm, ms in ctx.profile_end() if nm == "second_order"]
            res.setdefault(name, []).append((float(np.median(r)), float(min(r))))
            # both orders in one call (vs_reduce_first_second_order): every kernel of the call
            acc = ctx.alloc(8 * _native.acc_len(k, 1, 1))
            for _ in range(3):
                ctx.reduce_first_second_order(f, n, k, 1, acc, gram)
            ctx.sync()
            ctx.profile_begin(256)
            for _ in range(10):
                ctx.reduce_first_second_order(f, n, k, 1, acc, gram)
            recs = ctx.profile_end()
            both.setdefault(name, []).append((sum(ms for nm, ms in recs) / 10.0,
                                              float(np.median([ms for nm, ms in recs if nm == "second_order"]))))
            acc.free()
            grams[name] = ctx.download(gram, (2, k, k))
            f.free(); gram.free()
    a, b = grams["gram_dma_kernel"], grams["gram_st_kernel"]
    gb = 16.0 * k * n / 1e9
    for name in ctxs:
        med = [m for m, _ in res[name]]
        print("k=%d n=2^%d %-16s median of 10 per round: %s ms  (best round %.4f ms = %.2f TB/s)" %
              (k, int(np.log2(n)), name, " ".join("%.4f" % m for m in med), min(med), gb / min(med)))
    for name in ctxs:
        print("k=%d n=2^%d %-16s both orders in one call, kernels of the call per round: %s ms (Gram kernel %s)" %
              (k, int(np.log2(n)), name, " ".join("%.4f" % w for w, _ in both[name]), " ".join("%.4f" % g for _, g in both[name])))
    print("  max relative difference of the two kernels' sums: %.2e" % float(np.max(np.abs(a - b)) / np.max(np.abs(a))))
