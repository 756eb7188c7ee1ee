"""Randomised shapes (hypothesis): the kernels against the oracle over many small (k, rows, l) combinations,
including the edges of tiles, passes and blocks (k around 32/64, rows around 64/512/2048, odd rows)."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import saltelli_oracle as orc
from varsens_amd.saltelli import sobol_directions

pytestmark = pytest.mark.gpu

_ctx = {}


def ctx():
    if "c" not in _ctx:
        from varsens_amd import _native
        _ctx["c"] = _native.Context(0)
    return _ctx["c"]


EDGE_ROWS = st.one_of(st.integers(1, 70), st.sampled_from([63, 64, 65, 127, 128, 129, 511, 512, 513, 2047, 2048, 2049, 4097]),
                      st.integers(71, 6000))
EDGE_K = st.one_of(st.integers(1, 12), st.sampled_from([15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 70, 80, 81, 96, 97, 112, 113, 127, 128]))
# None: a Halton design (two draws of three); else the first index of a Sobol design -- 64-aligned (the M_1 points of an
# aligned block share their upper Gray-code bytes), off by one, across byte boundaries of the Gray code, anywhere
SOBOL_FIRST = st.one_of(st.none(), st.none(), st.sampled_from([0, 1, 64, 4096, 4097, 65535, 65536, (1 << 24) - 64, 1 << 24]),
                        st.integers(0, 1 << 26))
# VS_PROPERTY_EXAMPLES=N: a soak run with N random examples per test instead of the fixed 150 / 100
_SOAK = int(os.environ.get("VS_PROPERTY_EXAMPLES", "0"))
COMMON = dict(deadline=None, max_examples=_SOAK or 150, suppress_health_check=list(HealthCheck), derandomize=not _SOAK,
              database=None)


@settings(**COMMON)
@given(k=EDGE_K, rows=EDGE_ROWS, l=st.integers(1, 3), seed=st.integers(0, 2 ** 20), offset=st.sampled_from([0.0, 1.0, -3.0, 1e4]))
def test_reduction_any_shape(k, rows, l, seed, offset):
    from varsens_amd import _native
    c = ctx()
    rng = np.random.RandomState(seed)
    blocks = offset + rng.randn(2 * k + 2, rows, l)
    f = c.upload_blocks(blocks)
    nacc = _native.acc_len(k, l, 1)
    acc = c.alloc(8 * nacc)
    c.reduce_first_order(f, rows, k, l, acc)
    rec = c.download(acc, (nacc,))
    per = _native.acc_per_output(k)
    body = rec[4:4 + l * per].reshape(l, per)
    a, b, C, D = blocks[0], blocks[1], blocks[2:2 + k], blocks[2 + k:]
    scale = max(1.0, float(np.abs(blocks).max()) ** 2 * rows)
    np.testing.assert_allclose(body[:, 0], (a * b).sum(axis=0), rtol=1e-11, atol=1e-13 * scale)
    np.testing.assert_allclose(body[:, 3:3 + k].T, (a[None] * C).sum(axis=1), rtol=1e-11, atol=1e-13 * scale)
    np.testing.assert_allclose(body[:, 3 + k:3 + 2 * k].T, (b[None] * D).sum(axis=1), rtol=1e-11, atol=1e-13 * scale)
    np.testing.assert_allclose(body[:, 3 + 2 * k:3 + 3 * k].T, (a[None] * D).sum(axis=1), rtol=1e-11, atol=1e-13 * scale)
    np.testing.assert_allclose(body[:, 3 + 3 * k:3 + 4 * k].T, (b[None] * C).sum(axis=1), rtol=1e-11, atol=1e-13 * scale)
    if rows >= 2:
        out = c.alloc(8 * _native.first_order_len(k, l))
        c.finish_first_order(acc, rows, k, l, out)
        fin = c.download(out, (2 + 4 * k, l))
        np.testing.assert_allclose(fin[1], np.var(np.concatenate([a, b]), axis=0, ddof=1), rtol=1e-9)
        K = k * l
        gram = c.alloc(8 * 2 * K * K)
        c.reduce_second_order(f, rows, k, l, gram)
        G = c.download(gram, (2, K, K))
        Cm = C.transpose(0, 2, 1).reshape(K, rows); Dm = D.transpose(0, 2, 1).reshape(K, rows)
        np.testing.assert_allclose(G[0], (Cm + Dm) @ (Cm + Dm).T, rtol=1e-10, atol=4e-12 * scale)
        np.testing.assert_allclose(G[1], (Cm - Dm) @ (Cm - Dm).T, rtol=1e-10, atol=4e-12 * scale)
        gram.free(); out.free()
    f.free(); acc.free()


GRAM_K = st.one_of(st.integers(13, 34), st.integers(33, 64), st.sampled_from([40, 48, 63, 64, 65, 100, 127, 128, 129, 130]))


@settings(**dict(COMMON, max_examples=_SOAK or 100))
@given(k=GRAM_K, rows=EDGE_ROWS, seed=st.integers(0, 2 ** 20), trim=st.booleans())
def test_both_orders_any_shape(k, rows, seed, trim):
    """vs_reduce_first_second_order (one output) against the two separate passes and against numpy, over the
    staging variants the row count and k select: 16- and 32-sample-row units, wave-private and macro kernels,
    ragged ends, the NaN-masked second pass."""
    from varsens_amd import _native
    c = ctx()
    rng = np.random.RandomState(seed)
    blocks = 1.5 + rng.randn(2 * k + 2, rows, 1)
    flags = 0
    keep = np.ones(rows, bool)
    if trim and rows >= 8:
        bad = rng.choice(rows, 3, replace=False)
        blocks[rng.randint(0, 2 * k + 2, 3), bad, 0] = np.nan
        keep[bad] = False
        flags = _native.REDUCE_NAN_TRIM
    f = c.upload_blocks(blocks)
    nacc = _native.acc_len(k, 1, 1)
    acc1, acc2 = c.alloc(8 * nacc), c.alloc(8 * nacc)
    g1, g2 = c.alloc(8 * 2 * k * k), c.alloc(8 * 2 * k * k)
    c.reduce_first_order(f, rows, k, 1, acc1, flags=flags)
    c.reduce_second_order(f, rows, k, 1, g1, flags=flags)
    c.reduce_first_second_order(f, rows, k, 1, acc2, g2, flags=flags)
    a1, a2 = c.download(acc1, (nacc,)), c.download(acc2, (nacc,))
    G1, G2 = c.download(g1, (2, k, k)), c.download(g2, (2, k, k))
    Cm, Dm = blocks[2:2 + k, keep, 0], blocks[2 + k:, keep, 0]
    scale = max(1.0, float(np.abs(Cm).max()) ** 2 * rows) if keep.any() else 1.0
    np.testing.assert_allclose(G1[0], (Cm + Dm) @ (Cm + Dm).T, rtol=1e-10, atol=4e-12 * scale)
    np.testing.assert_allclose(G1[1], (Cm - Dm) @ (Cm - Dm).T, rtol=1e-10, atol=4e-12 * scale)
    np.testing.assert_allclose(G2, G1, rtol=1e-12, atol=1e-13 * scale)
    assert np.array_equal(a1[:4], a2[:4])
    np.testing.assert_allclose(a2, a1, rtol=1e-12, atol=1e-12 * scale)
    for b in (f, acc1, acc2, g1, g2):
        b.free()


@settings(**COMMON)
@given(k=EDGE_K, n=st.one_of(st.integers(1, 200), st.sampled_from([63, 64, 65, 128, 1000])),
       kind=st.sampled_from(["g", "lin", "poly", "inv"]), l=st.integers(1, 2), discard=st.integers(0, 50),
       begin_frac=st.floats(0, 1), mode=st.sampled_from([0, 1]), sobol_first=SOBOL_FIRST)
def test_evaluation_any_shape(k, n, kind, l, discard, begin_frac, mode, sobol_first):
    from varsens_amd import _native
    c = ctx()
    rng = np.random.RandomState(k * 1000 + n)
    lo = rng.rand(k) - 0.5
    hi = lo + 0.5 + rng.rand(k)
    perm = orc.seed1_permutation(n).astype(np.uint32)
    scaling = (lambda x: orc.scale_linear(x, lo, hi)) if mode else None
    if sobol_first is None:
        d = c.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k, discard), mode, lo if mode else None,
                            hi if mode else None, perm)
        S = orc.build_sample(k, n, scaling, discard=discard)
    else:       # a Sobol design from any first index: aligned blocks take the one-look-up M_1 path, others the general one
        dirs = sobol_directions(k)
        d = c.design_create(_native.GEN_SOBOL, k, n, sobol_first, mode, lo if mode else None, hi if mode else None, perm,
                            None, dirs)
        S = orc.build_sample(k, n, scaling, raw=orc.sobol(k, sobol_first, 2 * n, dirs))
    if kind == "inv":
        lo_, params, code = k, None, _native.OBJ_INVERT
        rows_func = lambda X: 1.0 - X
    else:
        lo_ = l
        P = 0.5 + rng.rand(l, k)
        params = P
        code = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[kind]
        rows_func = None
    if rows_func is None:     # separable built-ins: the engine's default (scan) association order
        want = orc.evaluate_engine_order({"g": "g_function", "lin": "linear", "poly": "polynomial"}[kind], P, S[0], S[1])
    else:
        want = orc.evaluate_rows(rows_func, *S)
    want = np.concatenate([want[0][None], want[1][None], want[2], want[3]], axis=0)
    begin = int(begin_frac * (n - 1))
    count = n - begin
    obj = c.objective_create(code, k, lo_, params)
    f = c.alloc(8 * 2 * (k + 1) * count * lo_)
    c.eval(d, obj, begin, count, f)
    got = c.download_blocks(f, k, lo_, count)
    assert np.array_equal(got, want[:, begin:begin + count])
    f.free(); obj.free(); d.free()


FUSED_K = st.one_of(st.integers(1, 128), st.sampled_from([16, 17, 32, 33, 48, 49, 64, 65, 80, 81, 96, 97, 112, 113, 128]))


def fused_both_takes(k, l, mode):
    """include/varsens_hip.h, vs_eval_reduce_first_second_order: several outputs of a built-in separable objective --
    2 .. 8 outputs up to 32 index rows, k >= 16 up to 64 (not with power scaling), 2 or 4 outputs up to 128 (3 up to 112)"""
    if l == 1:
        return k <= 128
    K = k * l
    return ((K <= 32 and l <= 8) or (K <= 64 and k >= 16 and mode != 2) or (64 < K <= 128 and (l in (2, 4) or (l == 3 and K <= 112))))


@settings(**dict(COMMON, max_examples=_SOAK or 160))
@given(k=FUSED_K, n=st.one_of(st.integers(1, 300), st.sampled_from([63, 64, 65, 127, 128, 129, 1000, 4097])),
       kind=st.sampled_from(["g", "lin", "poly"]), mode=st.sampled_from([0, 1, 2]), begin_frac=st.floats(0, 1),
       discard=st.integers(0, 40), slot=st.integers(0, 2), l=st.sampled_from([1, 1, 2, 3, 4]), sobol_first=SOBOL_FIRST)
def test_fused_both_orders_any_shape(k, n, kind, mode, begin_frac, discard, slot, l, sobol_first):
    """vs_eval_reduce_first_second_order == vs_eval + vs_reduce_first_second_order on any window of the design's rows,
    every kernel family (four waves x 4 / 8 columns, four and eight waves x 16; with several outputs four waves x 2 / 4 / 8
    and eight x 8 columns), every scaling, any variance slot, one to four outputs."""
    from varsens_amd import _native
    if l > 1:
        k = max(1, min(k, 128 // l))
    c = ctx()
    rng = np.random.RandomState(k * 977 + n + l)
    lo = hi = None
    if mode == 1:
        lo = rng.rand(k) - 0.5
        hi = lo + 0.5 + rng.rand(k)
    elif mode == 2:
        lo, hi = np.full(k, 1e-2), 0.5 + 0.5 * rng.rand(k)
    perm = orc.seed1_permutation(n).astype(np.uint32)
    if sobol_first is None:
        d = c.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k, discard), mode, lo, hi, perm)
    else:
        d = c.design_create(_native.GEN_SOBOL, k, n, sobol_first, mode, lo, hi, perm, None, sobol_directions(k))
    params = 0.5 + rng.rand(l, k)
    code = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[kind]
    obj = c.objective_create(code, k, l, params)
    takes = fused_both_takes(k, l, mode)
    assert c.eval_reduce_both_supported(d, obj) == takes
    if not takes:
        obj.free(); d.free()
        return
    K = k * l
    begin = int(begin_frac * (n - 1))
    count = n - begin
    ranks = 3
    nacc, ng = _native.acc_len(k, l, ranks), _native.gram_len(K)
    f = c.alloc(8 * 2 * (k + 1) * count * l)
    c.eval(d, obj, begin, count, f)
    acc1, acc2, g1, g2 = c.alloc(8 * nacc), c.alloc(8 * nacc), c.alloc(8 * ng), c.alloc(8 * ng)
    c.reduce_first_second_order(f, count, k, l, acc1, g1, 0, slot, ranks)
    c.eval_reduce_first_second_order(d, obj, begin, count, acc2, g2, slot, ranks)
    r1, r2 = c.download(acc1, (nacc,)), c.download(acc2, (nacc,))
    G1, G2 = c.download(g1, (2, K, K)), c.download(g2, (2, K, K))
    per = _native.acc_per_output(k)
    scale = max(1.0, float(np.abs(r1[4:4 + l * per]).max()))
    np.testing.assert_allclose(r2[:4 + l * per], r1[:4 + l * per], rtol=1e-11, atol=1e-12 * scale)
    # variance slots (count, mean[l], m2[l]): count exact, mean and m2 to rounding (m2 is a difference of sums: absolute
    # tolerance from its scale)
    s1, s2 = r1[4 + l * per:].reshape(ranks, 1 + 2 * l), r2[4 + l * per:].reshape(ranks, 1 + 2 * l)
    assert np.array_equal(s1[:, 0], s2[:, 0])
    np.testing.assert_allclose(s2[:, 1:1 + l], s1[:, 1:1 + l], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(s2[:, 1 + l:], s1[:, 1 + l:], rtol=1e-7, atol=1e-9 * scale)
    np.testing.assert_allclose(G2, G1, rtol=1e-11, atol=1e-12 * max(1.0, float(np.abs(G1).max())))
    for b in (f, acc1, acc2, g1, g2, obj, d):
        b.free()
