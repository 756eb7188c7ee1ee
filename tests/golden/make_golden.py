#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF in the build container.

Run from the repo root, only where /root/reference exists:

    python tests/golden/make_golden.py

How: the reference is Python 2 (print statements), so ``varsens/saltelli.py``
and ``varsens/scale.py`` are read as text, passed through the standard
library's ``lib2to3`` IN MEMORY and exec'd into fresh module objects.  Nothing
converted is written anywhere; only numeric inputs/outputs go into the .npz
files.  The reference imports the third-party ``ghalton`` module, which is not
installed; a stand-in is registered whose ``Halton(k).get(m)`` hands out this
build's own Halton definition (oracle ``halton``), first point index 1.  Halton
VALUES are therefore not pinned by these fixtures (see the oracle's header);
everything the reference computes from them is.

The reference's one-shot constructor crashes at HEAD (positional-argument
drift, varsens/saltelli.py:555,567), so every case goes through the 3-step
route Sample -> Objective -> Varsens(objective) with keyword arguments.
"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import saltelli_oracle as orc  # noqa: E402

REF = os.environ.get("VARSENS_REFERENCE", "/root/reference")


def load_reference():
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))

    class _Halton(object):
        def __init__(self, k):
            self.k = int(k)
            self.next = 1

        def get(self, m):
            pts = orc.halton(self.k, self.next, int(m))
            self.next += int(m)
            return pts.tolist()

    gh = types.ModuleType("ghalton")
    gh.Halton = _Halton
    sys.modules["ghalton"] = gh

    mods = {}
    for name in ("scale", "saltelli"):
        path = os.path.join(REF, "varsens", name + ".py")
        with open(path) as fh:
            src = fh.read() + "\n"
        code = compile(str(tool.refactor_string(src, path)), path, "exec")
        mod = types.ModuleType("ref_" + name)
        exec(code, mod.__dict__)
        mods[name] = mod
    return mods["saltelli"], mods["scale"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run3(ref, k, n, scaling, func, raw=None):
    s = ref.Sample(k, n, scaling, verbose=False, raw=raw)
    o = ref.Objective(k, n, s, func, verbose=False)
    v = ref.Varsens(o, verbose=False)
    return s, o, v


def result_fields(v):
    return dict(E_2=np.asarray(v.E_2), var_y=np.asarray(v.var_y), U_j=v.U_j, U_nj=v.U_nj,
                sens=v.sens, sens_t=v.sens_t, sens_2=v.sens_2, sens_2n=v.sens_2n)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def main():
    ref, rscale = load_reference()
    ident = lambda x: x  # noqa: E731

    # ---- G1/G2: Sobol g-function, a=[0,.5,3,9,99,99], k=6 (test_g_function.py:18,54)
    a6 = [0, 0.5, 3, 9, 99, 99]

    def gi(xi, ai):
        return (np.abs(4.0 * xi - 2.0) + ai) / (1.0 + ai)

    def g_obj(x):
        return np.prod([gi(xi, a6[i]) for i, xi in enumerate(x)])

    def g_obj2(x):
        return [np.prod([gi(xi, a6[i]) for i, xi in enumerate(x)]),
                np.prod([gi(xi, a6[::-1][i]) for i, xi in enumerate(x)])]

    for tag, n in (("g1_gfunc_k6_n1024", 1024), ("g2_gfunc_k6_n51200", 51200)):
        s, o, v = run3(ref, 6, n, ident, g_obj)
        save(tag, k=6, n=n, a=np.array(a6, dtype=float),
             M_1_head=s.M_1[:8], M_2_head=s.M_2[:8], flat_sha=sha(s.flat()),
             fM_1_head=o.fM_1[:8], fN_j_head=o.fN_j[:, :4], fN_nj_head=o.fN_nj[:, :4],
             fvals_sha=sha(o.flat()), **result_fields(v))

    # ---- G3: two outputs (test_g_function.py:77-88)
    s, o, v = run3(ref, 6, 1024, ident, g_obj2)
    save("g3_gfunc2_k6_n1024", k=6, n=1024, a=np.array(a6, dtype=float),
         fvals_sha=sha(o.flat()), **result_fields(v))

    # ---- G4: Ishigami, linear scaling to [-pi, pi]
    lo = np.array([-np.pi] * 3)
    hi = np.array([np.pi] * 3)

    def ishi(x):
        return np.sin(x[0]) + 7.0 * np.sin(x[1]) ** 2 + 0.1 * x[2] ** 4 * np.sin(x[0])
    s, o, v = run3(ref, 3, 4096, lambda x: rscale.linear(x, lo, hi), ishi)
    save("g4_ishigami_k3_n4096", k=3, n=4096, lower=lo, upper=hi,
         M_1_head=s.M_1[:8], M_2_head=s.M_2[:8], fM_1_head=o.fM_1[:8], **result_fields(v))

    # ---- G5: linear objective k=32, c_i = i+1
    c32 = np.arange(1, 33, dtype=float)
    s, o, v = run3(ref, 32, 4096, ident, lambda x: float(np.dot(c32, x)))
    save("g5_linear_k32_n4096", k=32, n=4096, c=c32, fM_1_head=o.fM_1[:8], **result_fields(v))

    # ---- G5b: separable polynomial k=16, beta_i = 1/(1+i/8)
    beta = 1.0 / (1.0 + np.arange(16) / 8.0)
    s, o, v = run3(ref, 16, 2048, ident, lambda x: float(np.prod(1.0 + beta * (x - 0.5))))
    save("g5b_poly_k16_n2048", k=16, n=2048, beta=beta, fM_1_head=o.fM_1[:8], **result_fields(v))

    # ---- G6: NaN trimming on the pre-computed-values route (saltelli.py:474-495)
    rng = np.random.RandomState(20260101)
    k, n, l = 5, 64, 2
    vals = 1.0 + rng.rand(2 * n * (k + 1), l)
    nan_at = [(3, 0), (n + 3, 0), (2 * n + 17, 0), ((2 + k) * n + 4 * n + 40, 0), (9, 1), (n + 50, 1)]
    for r, c in nan_at:
        vals[r, c] = np.nan
    o = ref.Objective(k, n, objective_vals=vals.copy(), verbose=False)
    v = ref.Varsens(o, verbose=False)
    save("g6_nantrim_k5_n64_l2", k=k, n=n, l=l, vals=vals, kept=o.fM_1.shape[0],
         fM_1=o.fM_1, fN_nj=o.fN_nj, **result_fields(v))

    # ---- G7: the seed-1 permutation (saltelli.py:100-101), via a marker column
    perms = {}
    for n in (1, 2, 5, 13, 1024, 100003):
        raw = np.tile((np.arange(2 * n, dtype=float) / (2 * n))[:, None], (1, 2))
        s = ref.Sample(2, n, ident, verbose=False, raw=raw)
        p = np.rint(s.M_2[:, 0] * (2 * n)).astype(np.int64) - n
        assert sorted(p.tolist()) == list(range(n))
        perms["n%d" % n] = p if n <= 1024 else np.concatenate([p[:64], p[-64:]])
        perms["sha_n%d" % n] = sha(p)
    save("g7_seed1_perm", **perms)

    # ---- G8: Sample.flat() with power scaling on an injected raw sample (test_sample.py:66-75 shape)
    rng = np.random.RandomState(8)
    k, n = 5, 13
    raw = rng.rand(2 * n, k)
    plo = np.array([1.0, 10.0, 1e-3, 5.0, 100.0])
    phi = np.array([100.0, 20.0, 1e3, 6.0, 2500.0])
    s = ref.Sample(k, n, lambda x: rscale.power(x, plo, phi), verbose=False, raw=raw)
    save("g8_flat_power_k5_n13", k=k, n=n, raw=raw, lower=plo, upper=phi, flat=s.flat(),
         M_1=s.M_1, M_2=s.M_2, N_j=s.N_j, N_nj=s.N_nj)

    # ---- G9: scale.py known answers (test_scaling.py + doctests), computed by the reference
    pts = np.arange(5) / 4.0
    refv = np.array([1, 10, 100, 1000, 1e4])
    save("g9_scale",
         pts=pts,
         linear=rscale.linear(pts, np.array([-100.0] * 5), np.array([100.0] * 5)),
         power=rscale.power(pts, np.array([1.0] * 5), np.array([100.0] * 5)),
         percentage=rscale.percentage(pts, np.array([-10.0, -1, 0, 1, 20]), 33.0),
         magnitude=rscale.magnitude(pts, refv),
         doc_linear=rscale.linear(np.array([0.5] * 3), np.array([-100, -10, 1000]), np.array([100, 20, 2000])),
         doc_power=rscale.power(np.array([0.5] * 3), np.array([10, 100, 1000]), np.array([1000, 200, 2500])),
         doc_percentage=rscale.percentage(np.array([0.333] * 3), np.array([1, 10, 1000]), 50.0),
         doc_magnitude=rscale.magnitude(np.array([0.333] * 3), np.array([1, 10, 1000])))

    # ---- G10: percentage / magnitude scaling through Sample(raw=...)
    rng = np.random.RandomState(10)
    k, n = 4, 7
    raw = rng.rand(2 * n, k)
    refp = np.array([2.0, -3.0, 50.0, 0.25])
    s1 = ref.Sample(k, n, lambda x: rscale.percentage(x, refp, 20.0), verbose=False, raw=raw)
    s2 = ref.Sample(k, n, lambda x: rscale.magnitude(x, np.abs(refp), 2.0, 10.0), verbose=False, raw=raw)
    save("g10_pct_mag_k4_n7", k=k, n=n, raw=raw, reference=refp,
         pct_M_1=s1.M_1, pct_M_2=s1.M_2, pct_N_j=s1.N_j, pct_N_nj=s1.N_nj,
         mag_M_1=s2.M_1, mag_M_2=s2.M_2, mag_flat=s2.flat())

    # ---- G11: vector objective invert(x)=1-x, l=k (test_objective.py:5,18-42)
    k, n = 8, 23
    s = ref.Sample(k, n, ident, verbose=False)
    o = ref.Objective(k, n, s, lambda x: 1.0 - x, verbose=False)
    v = ref.Varsens(o, verbose=False)
    save("g11_invert_k8_n23", k=k, n=n, M_1=s.M_1, M_2=s.M_2, fM_1=o.fM_1, fM_2=o.fM_2,
         fN_j=o.fN_j, fN_nj=o.fN_nj, flat_obj=o.flat(), **result_fields(v))

    # ---- G12: pre-computed values, l=1 and l=3 (saltelli.py:449-470), incl. /scaling
    rng = np.random.RandomState(12)
    k, n = 7, 50
    for l in (1, 3):
        vals = 3.0 + rng.randn(2 * n * (k + 1), l)
        o = ref.Objective(k, n, objective_vals=vals.copy(), verbose=False)
        v = ref.Varsens(o, verbose=False)
        save("g12_vals_k7_n50_l%d" % l, k=k, n=n, l=l, vals=vals, **result_fields(v))
    o = ref.Objective(k, n, verbose=False, obj_vals=vals.copy(), scaling=4.0)  # -> load(**loadArgs)
    v = ref.Varsens(o, verbose=False)
    save("g12_vals_k7_n50_l3_div4", k=k, n=n, l=3, vals=vals, divide=4.0, **result_fields(v))

    # ---- G13: discard continues the stream (saltelli.py:28-33,83)
    s = ref.Sample(3, 16, ident, 37, verbose=False)
    save("g13_discard37_k3_n16", k=3, n=16, discard=37, M_1=s.M_1, M_2=s.M_2)


def wire_formats():
    """G14: the text wire formats (Sample.export / Objective.export, saltelli.py:173-193,393-413):
    file names and exact bytes written by the reference for a tiny design."""
    import tempfile
    ref, rscale = load_reference()
    k, n = 3, 4
    s = ref.Sample(k, n, lambda x: x, verbose=False)
    o = ref.Objective(k, n, s, lambda x: [float(x[0] + 2 * x[1]), float(x[2] * x[2])], verbose=False)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        s.export(d, "my sample", ".csv", 10, ",")           # 32 rows -> 4 files
        s.export(d, "whole", ".txt")                        # one file, tab separated
        o.export(d, "obj", ".dat", 20)                      # 32 rows -> 2 files
        names = sorted(os.listdir(d))
        out["names"] = np.array(names)
        for name in names:
            with open(os.path.join(d, name), "rb") as fh:
                out["file_" + name.replace(".", "_")] = np.frombuffer(fh.read(), dtype=np.uint8)
    save("g14_wire_formats_k3_n4", k=k, n=n, **out)


def nan_fm_only():
    """G6b: the only NaNs sit in fM_1 / fM_2 rows (blocks the second-order products never read);
    the reference still trims those rows from fN_j / fN_nj (saltelli.py:474-492)."""
    ref, rscale = load_reference()
    rng = np.random.RandomState(20260102)
    k, n, l = 4, 96, 1
    vals = 1.0 + rng.rand(2 * n * (k + 1), l)
    vals[11, 0] = np.nan          # fM_1 row 11
    vals[n + 70, 0] = np.nan      # fM_2 row 70
    o = ref.Objective(k, n, objective_vals=vals.copy(), verbose=False)
    v = ref.Varsens(o, verbose=False)
    save("g6b_nantrim_fm_only_k4_n96", k=k, n=n, l=l, vals=vals, kept=o.fM_1.shape[0], **result_fields(v))


def config2():
    """G15: BASELINE configs[1] -- Sobol g-function, k=6, n=2^20 -- through the reference's own
    3-step route (14.7 M calls of the objective: a few minutes in the build container).  Only the
    results go into the fixture, plus checksums / a head of the f-values the reference produced."""
    ref, rscale = load_reference()
    a6 = [0, 0.5, 3, 9, 99, 99]
    arr = np.array(a6, dtype=float)

    def g_obj(x):  # the same arithmetic as g_obj in main(): per-factor terms, numpy.prod over 6 values
        return np.prod((np.abs(4.0 * x - 2.0) + arr) / (1.0 + arr))

    n = 1 << 20
    s, o, v = run3(ref, 6, n, lambda x: x, g_obj)
    save("g15_gfunc_k6_n1048576", k=6, n=n, a=arr,
         M_1_head=s.M_1[:8], M_2_head=s.M_2[:8],
         fM_1_head=o.fM_1[:8], fM_2_head=o.fM_2[:8], fN_j_head=o.fN_j[:, :4], fN_nj_head=o.fN_nj[:, :4],
         fvals_sha=sha(o.flat()), **result_fields(v))


def several_outputs():
    """G16, G17 (round 4): models with several outputs at the sizes of the one-kernel default call's several-output kernels,
    run by the reference itself -- its objective returns a list per row (varsens/saltelli.py:329-355), compute_varsens
    gives sens_2[k, l, k, l] (:612-622).
    G16: linear, k = 24, three outputs (72 index rows: the eight-wave kernel), scale.linear(-1, 2), n = 2048;
    G17: g-function, k = 16, two outputs (32 index rows: a four-wave kernel), identity, n = 2048."""
    ref, rscale = load_reference()
    k, l, n = 24, 3, 2048
    C = np.arange(1, k + 1, dtype=float)[None, :] * (1.0 + np.arange(l))[:, None] + np.arange(l)[:, None] ** 2
    lo, hi = np.full(k, -1.0), np.full(k, 2.0)
    s, o, v = run3(ref, k, n, lambda x: rscale.linear(x, lo, hi), lambda x: [float(np.dot(c, x)) for c in C])
    save("g16_linear3_k24_n2048", k=k, n=n, l=l, c=C, lower=lo, upper=hi, fM_1_head=o.fM_1[:8], **result_fields(v))
    k, l, n = 16, 2, 2048
    A = np.stack([np.linspace(0.0, 9.0, k), np.linspace(9.0, 0.5, k)])

    def g2(x):
        return [float(np.prod((np.abs(4.0 * x - 2.0) + a) / (1.0 + a))) for a in A]
    s, o, v = run3(ref, k, n, lambda x: x, g2)
    save("g17_gfunc2_k16_n2048", k=k, n=n, l=l, a=A, fM_1_head=o.fM_1[:8], fvals_sha=sha(o.flat()), **result_fields(v))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not found at %s" % REF)
    if "--only-wire" in sys.argv:
        wire_formats()
    elif "--nan-fm" in sys.argv:
        nan_fm_only()
    elif "--config2" in sys.argv:   # minutes of reference time: generated on its own
        config2()
    elif "--several-outputs" in sys.argv:
        several_outputs()
    else:
        main()
        nan_fm_only()
        wire_formats()
        several_outputs()
