"""Parity of the HIP kernels with the oracle, through the C ABI (ctypes).  GPU only."""
import numpy as np
import pytest

from oracle import saltelli_oracle as orc

pytestmark = pytest.mark.gpu

A6 = [0, 0.5, 3, 9, 99, 99]


@pytest.fixture(scope="module")
def ctx():
    from varsens_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def same(a, b):
    assert a.shape == b.shape
    assert np.array_equal(a, b, equal_nan=True), "max |d| = %g" % np.nanmax(np.abs(a - b))


# ------------------------------------------------------------------ generators

# (the large starts: indices whose numerators the short digit step forms as i * b^(digits-1) close to 2^53 -- 3^17 - 1 is
# the last index with every column of k = 5 in that form --, indices where some columns are back on the digit loop, the
# last indices below 2^32; k = 1000: bases up to 7919, denominators up to 2^49 for the two-word quotient)
@pytest.mark.parametrize("k,start,count", [(1, 1, 1), (7, 1, 513), (40, 801, 5000), (128, 2561, 3000),
                                           (3, 2 ** 32 - 100, 300), (5, 3 ** 17 - 400, 400), (5, 3 ** 17 - 200, 400),
                                           (64, 2 ** 25 - 100, 300), (128, 2 ** 26 + 12345, 200), (1000, 60_000_000, 40),
                                           (16, 2 ** 27 - 50, 100), (32, 9 * 10 ** 8, 200)])
def test_halton_fill_bit_exact(ctx, k, start, count):
    buf = ctx.halton_fill(k, start, count)
    got = ctx.download(buf, (count, k))
    same(got, orc.halton(k, start, count))
    assert got.min() > 0.0 and got.max() < 1.0   # varsens/tests/test_sample.py:23-42


def test_halton_range_guard(ctx):
    from varsens_amd._native import VarsensHipError
    with pytest.raises(VarsensHipError):
        ctx.halton_fill(128, 2 ** 50, 10)


def test_sobol_fill_bit_exact(ctx):
    from varsens_amd.saltelli import sobol_directions
    d = sobol_directions(64)
    buf = ctx.sobol_fill(64, d, 4096, 5000)
    same(ctx.download(buf, (5000, 64)), orc.sobol(64, 4096, 5000, d))


def make_design(ctx, k, n, mode=0, lo=None, hi=None, raw=None, discard=0, perm=True):
    from varsens_amd import _native
    p = orc.seed1_permutation(n).astype(np.uint32) if perm else None
    if raw is None:
        return ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k, discard), mode, lo, hi, p)
    return ctx.design_create(_native.GEN_RAW, k, n, 0, mode, lo, hi, p, raw)


@pytest.mark.parametrize("k,n", [(5, 13), (3, 5), (11, 13), (17, 1024)])
def test_sample_flat_identity(ctx, k, n):
    d = make_design(ctx, k, n)
    rows = 2 * n * (k + 1)
    buf = ctx.alloc(8 * rows * k)
    ctx.sample_flat(d, 0, rows, buf)
    same(ctx.download(buf, (rows, k)), orc.flat_sample(*orc.build_sample(k, n)))


def test_sample_flat_linear_and_window(ctx):
    k, n = 6, 200
    lo = np.linspace(-3, 2, k); hi = lo + np.arange(1, k + 1)
    d = make_design(ctx, k, n, 1, lo, hi)
    want = orc.flat_sample(*orc.build_sample(k, n, lambda x: orc.scale_linear(x, lo, hi)))
    buf = ctx.alloc(8 * 777 * k)
    ctx.sample_flat(d, 1234, 777, buf)
    same(ctx.download(buf, (777, k)), want[1234:1234 + 777])


def test_sample_flat_power_raw(ctx, golden):
    g = golden("g8_flat_power_k5_n13")
    d = make_design(ctx, 5, 13, 2, g["lower"], g["upper"], raw=g["raw"])
    buf = ctx.alloc(8 * 156 * 5)
    ctx.sample_flat(d, 0, 156, buf)
    # lower * exp(x ln(upper / lower)) on the device vs the reference's numpy pow: one or two ulp
    np.testing.assert_allclose(ctx.download(buf, (156, 5)), g["flat"], rtol=1e-15)


def test_power_scaling_within_two_ulp_of_numpy(ctx):
    """scale.power (varsens/scale.py:62): lower * (upper / lower) ** x.  The device takes it as lower * exp(x ln r) with
    ln r a double-double from the host (round 4: one exp instead of a pow per point, 1.96 -> 1.05 ms at k = 32, n = 2^22);
    bounds over fifteen orders of magnitude, both negative, and -- ratio zero, bounds of opposite sign: no finite
    logarithm -- the pow() fallback with numpy's conventions."""
    from varsens_amd import _native
    k, n = 12, 20000
    rng = np.random.RandomState(3)
    lo = 10.0 ** rng.uniform(-6, 3, k)
    hi = lo * 10.0 ** rng.uniform(0.1, 9, k)
    lo[0], hi[0] = -3.0, -0.002          # both negative: the ratio is positive
    lo[1], hi[1] = 2.5, 0.0              # ratio 0: 0 ** x (pow fallback)
    lo[2], hi[2] = 4.0, 4.0              # ratio 1
    d = make_design(ctx, k, n, _native.SCALE_POWER, lo, hi)
    buf = ctx.alloc(8 * 2 * n * k)
    ctx.sample_flat(d, 0, 2 * n, buf)
    got = ctx.download(buf, (2 * n, k))
    with np.errstate(all="ignore"):
        S = orc.build_sample(k, n, lambda x: orc.scale_power(x, lo, hi))
    want = np.concatenate([S[0], S[1]])
    assert np.array_equal(got[:, 1], want[:, 1]) and np.array_equal(got[:, 2], want[:, 2])   # exact cases
    cols = [0] + list(range(3, k))
    ulp = np.abs(got[:, cols] - want[:, cols]) / (np.abs(want[:, cols]) * 2.0 ** -52)
    assert ulp.max() <= 2.5, ulp.max()
    assert ulp.mean() < 0.5


def test_power_scaling_overflow_is_inf_like_pow(ctx):
    """raw points at or above 1 with a ratio near DBL_MAX overflow exp(x ln r); the result must be pow's inf
    (0 for the underflow), not the NaN of fma(inf, e, inf)."""
    from varsens_amd import _native
    k, n = 3, 4
    lo = np.array([1.0, 2.0, 1e-3])
    hi = np.array([1.7e308, 1e308, 1e300])
    raw = np.array([[1.0, 1.5, 30.0], [0.999, 2.0, 1.25], [1.0000000001, 1.0, 0.5], [-1.5, -40.0, 1.0],
                    [0.0, 0.25, 1.0], [1.0, 1.0, 1.0], [3.0, 0.5, 2.0], [0.5, 0.75, -2.0]])
    d = make_design(ctx, k, n, _native.SCALE_POWER, lo, hi, raw=raw, perm=False)
    buf = ctx.alloc(8 * 2 * n * k)
    ctx.sample_flat(d, 0, 2 * n, buf)
    got = ctx.download(buf, (2 * n, k))
    with np.errstate(all="ignore"):
        want = lo * (hi / lo) ** raw
    assert not np.isnan(got).any()
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.isinf(want).sum() >= 5
    fin = np.isfinite(want) & (want != 0)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-13)
    assert np.array_equal(got[want == 0], want[want == 0])


# ------------------------------------------------------------------ fused evaluation

def eval_dev(ctx, design, kind, k, l, params, n, begin=0, count=None):
    count = n if count is None else count
    obj = ctx.objective_create(kind, k, l, params)
    f = ctx.alloc(8 * 2 * (k + 1) * count * l)
    ctx.eval(design, obj, begin, count, f)
    return f, ctx.download_blocks(f, k, l, count)


def oracle_blocks(F):
    fM_1, fM_2, fN_j, fN_nj = F
    return np.concatenate([fM_1[None], fM_2[None], fN_j, fN_nj], axis=0)


@pytest.mark.parametrize("k,n,l", [(6, 1024, 1), (6, 1024, 2), (1, 7, 1), (33, 300, 1), (70, 129, 1),
                                   (197, 700, 1), (100, 200, 2), (300, 70, 1)])
def test_eval_gfunction_bit_exact(ctx, k, n, l):
    """Up to 256 columns the scan kernels fold in the segmented order the oracle restates in evaluate_scan --
    bit-exact against that, rounding-close to the strict left-to-right product.  k = 300 exceeds both the
    scan kernels and the LDS tile: chain kernel (strict order) with its tile in an HBM scratch slice."""
    from varsens_amd import _native
    rng = np.random.RandomState(k)
    a = np.array(A6, dtype=float) if k == 6 else rng.rand(k) * 10
    A = np.stack([a, a[::-1]])[:l]
    d = make_design(ctx, k, n)
    _, got = eval_dev(ctx, d, _native.OBJ_G_FUNCTION, k, l, A, n)
    S = orc.build_sample(k, n)
    strict = oracle_blocks(orc.evaluate_engine_order("g_function", A, S[0], S[1], strict=True))
    same(got, oracle_blocks(orc.evaluate_engine_order("g_function", A, S[0], S[1])))
    assert np.max(np.abs(got - strict) / np.abs(strict)) <= 6e-15      # reassociation of <= 256 factors


# Halton designs around index 2^32.  The scan kernels keep 32-bit indices and per-column
# constant digit counts; vs_eval / vs_eval_reduce_first_order must leave them exactly when the LAST index of the
# design, first_index + 2n - 1, reaches 2^32 (vs_generate.hip design_compact) -- an off-by-one is a silent
# wrong answer.  discard moves the design (saltelli.py:28-33,83): first_index = 1 + 20k + discard.
_K32, _N32 = 5, 64
_EDGE = 2 ** 32 - 2 * _N32 - (1 + 20 * _K32)      # discard with first_index + 2n - 1 == 2^32 - 1
_DISCARDS_32 = [("last index 2^32-1: the scan kernels' last design", _EDGE, True),
                ("last index 2^32: first design beyond them", _EDGE + 1, False),
                ("2^32 inside M_2", _EDGE + 40, False),
                ("2^32 inside M_1", _EDGE + 100, False),
                ("all indices above 2^32", _EDGE + 300, False)]


@pytest.mark.parametrize("label,discard,compact", _DISCARDS_32, ids=[c[0] for c in _DISCARDS_32])
@pytest.mark.parametrize("scaled", [False, True], ids=["identity", "linear"])
def test_designs_around_index_2_pow_32(ctx, label, discard, compact, scaled):
    from varsens_amd import _native
    k, n = _K32, _N32
    first = orc.halton_skip(k, discard)
    assert (first + 2 * n - 1 < 2 ** 32) == compact
    lo, hi = np.linspace(-2.0, 1.0, k), np.linspace(3.0, 9.0, k)
    if scaled:
        d = make_design(ctx, k, n, _native.SCALE_LINEAR, lo, hi, discard=discard)
        S = orc.build_sample(k, n, lambda x: orc.scale_linear(x, lo, hi), discard=discard)
    else:
        d = make_design(ctx, k, n, discard=discard)
        S = orc.build_sample(k, n, discard=discard)
    # vs_sample_flat
    rows = 2 * n * (k + 1)
    buf = ctx.alloc(8 * rows * k)
    ctx.sample_flat(d, 0, rows, buf)
    same(ctx.download(buf, (rows, k)), orc.flat_sample(*S))
    # vs_eval: scan order while the design is compact, the strict chain kernels beyond
    a = np.array([0.0, 0.5, 3.0, 9.0, 99.0])
    for kind, name, params in ((_native.OBJ_G_FUNCTION, "g_function", a),
                               (_native.OBJ_LINEAR, "linear", np.arange(1.0, k + 1)),
                               (_native.OBJ_POLYNOMIAL, "polynomial", 1.0 / (1.0 + np.arange(k) / 8.0))):
        f, got = eval_dev(ctx, d, kind, k, 1, params, n)
        same(got, oracle_blocks(orc.evaluate_engine_order(name, params, S[0], S[1], strict=not compact)))
        # a window of rows (row_begin > 0) takes the same route
        _, part = eval_dev(ctx, d, kind, k, 1, params, n, begin=7, count=50)
        same(part, got[:, 7:57])
        # vs_eval_reduce_first_order: offered exactly while the scan kernels apply, and then equal to the
        # two-kernel path on the same design
        obj = ctx.objective_create(kind, k, 1, params)
        assert ctx.eval_reduce_supported(d, obj) == compact
        nacc = _native.acc_len(k, 1, 1)
        acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
        ctx.reduce_first_order(f, n, k, 1, acc1)
        r1 = ctx.download(acc1, (nacc,))
        ex = orc.exact_accumulators(*orc.evaluate_engine_order(name, params, S[0], S[1], strict=not compact))
        np.testing.assert_allclose(r1[4 + 3:4 + 3 + k], ex["ac"][:, 0], rtol=1e-13)
        if compact:
            ctx.eval_reduce_first_order(d, obj, 0, n, acc2)
            r2 = ctx.download(acc2, (nacc,))
            np.testing.assert_allclose(r2[4:4 + 3 + 4 * k], r1[4:4 + 3 + 4 * k], rtol=1e-13)
        else:
            with pytest.raises(_native.VarsensHipError):
                ctx.eval_reduce_first_order(d, obj, 0, n, acc2)


@pytest.mark.parametrize("label,discard,compact", _DISCARDS_32, ids=[c[0] for c in _DISCARDS_32])
def test_ishigami_around_index_2_pow_32(ctx, label, discard, compact):
    """Ishigami has its own thread-per-row kernels (evaluation and fused), offered for every design: they must
    carry 64-bit Halton indices themselves."""
    from varsens_amd import _native
    k, n = 3, _N32
    discard += 20 * (_K32 - k)                     # same first index as the k = 5 designs above
    lo, hi = np.full(k, -np.pi), np.full(k, np.pi)
    d = make_design(ctx, k, n, _native.SCALE_LINEAR, lo, hi, discard=discard)
    S = orc.build_sample(k, n, lambda x: orc.scale_linear(x, lo, hi), discard=discard)
    f, got = eval_dev(ctx, d, _native.OBJ_ISHIGAMI, k, 1, [7.0, 0.1], n)
    want = oracle_blocks(orc.evaluate_rows(lambda X: orc.ishigami_rows(X, 7.0, 0.1), *S))
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-14)          # device sin: a few ulp
    obj = ctx.objective_create(_native.OBJ_ISHIGAMI, k, 1, [7.0, 0.1])
    assert ctx.eval_reduce_supported(d, obj)
    nacc = _native.acc_len(k, 1, 1)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    ctx.reduce_first_order(f, n, k, 1, acc1)
    ctx.eval_reduce_first_order(d, obj, 0, n, acc2)
    r1, r2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    np.testing.assert_allclose(r2[4:4 + 3 + 4 * k], r1[4:4 + 3 + 4 * k], rtol=1e-12, atol=1e-12)


@pytest.fixture(scope="module")
def strict_ctx():
    """A context in strict evaluation order: separable objectives fold left to right (chain kernels), the
    bits of numpy.prod over the factors, which is what the reference's test objectives compute."""
    from varsens_amd import _native
    c = _native.Context(0)
    c.set_eval_order("strict")
    yield c
    c.close()


@pytest.mark.parametrize("k,n,l", [(6, 1024, 1), (6, 1024, 2), (1, 7, 1), (33, 300, 1), (70, 129, 1), (100, 200, 2)])
def test_eval_strict_order_bit_exact(strict_ctx, k, n, l):
    from varsens_amd import _native
    rng = np.random.RandomState(k)
    a = np.array(A6, dtype=float) if k == 6 else rng.rand(k) * 10
    A = np.stack([a, a[::-1]])[:l]
    _, got = eval_dev(strict_ctx, make_design(strict_ctx, k, n), _native.OBJ_G_FUNCTION, k, l, A, n)
    S = orc.build_sample(k, n)
    same(got, oracle_blocks(orc.evaluate_rows(
        lambda X: np.stack([orc.g_function_rows(X, A[o]) for o in range(l)], axis=-1), *S)))
    d = make_design(strict_ctx, k, n)
    obj = strict_ctx.objective_create(_native.OBJ_G_FUNCTION, k, l, A)
    assert not strict_ctx.eval_reduce_supported(d, obj)      # the fused kernels fold in scan order


def tuned_context(monkeypatch, **env):
    """A second context on GPU 0 created under measurement knobs (the engine reads them once, at
    vs_context_create)."""
    from varsens_amd import _native
    monkeypatch.setenv("VS_TUNING", "1")   # the knobs count only under this switch
    for name, value in env.items():
        monkeypatch.setenv(name, str(value))
    c = _native.Context(0)
    for name in list(env) + ["VS_TUNING"]:
        monkeypatch.delenv(name)
    return c


def test_tuning_knobs_need_the_switch_and_a_valid_value(ctx, monkeypatch, capfd):
    """A stray VS_* variable in a user's shell must not change kernels: without VS_TUNING=1 the knobs
    are ignored, with it a value outside the knob's range keeps the default; both are said on stderr."""
    from varsens_amd import _native
    k, n = 12, 4096
    a = np.linspace(0.1, 9.0, k)
    _, want = eval_dev(ctx, make_design(ctx, k, n), _native.OBJ_G_FUNCTION, k, 1, a, n)
    monkeypatch.setenv("VS_EVAL_MODE", "1")          # would select the strict-order kernels
    monkeypatch.setenv("VS_SCAN_SHAPE", "777")       # not an instantiated shape: would make vs_eval fail
    c2 = _native.Context(0)
    _, got = eval_dev(c2, make_design(c2, k, n), _native.OBJ_G_FUNCTION, k, 1, a, n)
    same(got, want)
    c2.close()
    monkeypatch.setenv("VS_TUNING", "1")
    monkeypatch.setenv("VS_EVAL_MODE", "9")
    monkeypatch.delenv("VS_SCAN_SHAPE")
    c3 = _native.Context(0)
    _, got = eval_dev(c3, make_design(c3, k, n), _native.OBJ_G_FUNCTION, k, 1, a, n)
    same(got, want)
    c3.close()
    err = capfd.readouterr().err
    assert "VS_EVAL_MODE=9 is outside" in err


def test_eval_scratch_tile_path_forced(strict_ctx, monkeypatch):
    """The HBM-scratch tile variant of the chain kernel on a size that also fits LDS: identical bits, several
    blocks per workgroup."""
    from varsens_amd import _native
    k, n = 12, 300000
    a = np.linspace(0.1, 9.0, k)
    _, want = eval_dev(strict_ctx, make_design(strict_ctx, k, n), _native.OBJ_G_FUNCTION, k, 1, a, n)
    c2 = tuned_context(monkeypatch, VS_EVAL_SCRATCH=1)
    d2 = make_design(c2, k, n)
    _, got = eval_dev(c2, d2, _native.OBJ_G_FUNCTION, k, 1, a, n)
    same(got, want)
    _, got = eval_dev(c2, d2, _native.OBJ_INVERT, k, k, None, 5000)
    _, want = eval_dev(strict_ctx, make_design(strict_ctx, k, n), _native.OBJ_INVERT, k, k, None, 5000)
    same(got, want)
    d2.free()
    c2.close()


def test_eval_linear_polynomial_bit_exact(ctx):
    from varsens_amd import _native
    k, n = 32, 500
    d = make_design(ctx, k, n)
    S = orc.build_sample(k, n)
    c = np.arange(1, k + 1, dtype=float)
    _, got = eval_dev(ctx, d, _native.OBJ_LINEAR, k, 1, c, n)
    same(got, oracle_blocks(orc.evaluate_engine_order("linear", c, S[0], S[1])))
    beta = 1.0 / (1.0 + np.arange(k) / 8.0)
    _, got = eval_dev(ctx, d, _native.OBJ_POLYNOMIAL, k, 1, beta, n)
    same(got, oracle_blocks(orc.evaluate_engine_order("polynomial", beta, S[0], S[1])))


def test_eval_ishigami(ctx):
    from varsens_amd import _native
    n = 4096
    lo = np.full(3, -np.pi); hi = np.full(3, np.pi)
    d = make_design(ctx, 3, n, 1, lo, hi)
    _, got = eval_dev(ctx, d, _native.OBJ_ISHIGAMI, 3, 1, [7.0, 0.1], n)
    S = orc.build_sample(3, n, lambda x: orc.scale_linear(x, lo, hi))
    want = oracle_blocks(orc.evaluate_rows(orc.ishigami_rows, *S))
    # device sin vs libm sin: <= 2 ulp of values up to ~18
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)


def test_eval_invert_and_shard(ctx, golden):
    from varsens_amd import _native
    g = golden("g11_invert_k8_n23")
    d = make_design(ctx, 8, 23)
    _, got = eval_dev(ctx, d, _native.OBJ_INVERT, 8, 8, None, 23)
    same(got[0], g["fM_1"]); same(got[1], g["fM_2"]); same(got[2:10], g["fN_j"]); same(got[10:], g["fN_nj"])
    # a shard of rows evaluates to the same values (multi-GPU row split)
    _, part = eval_dev(ctx, d, _native.OBJ_INVERT, 8, 8, None, 23, begin=5, count=11)
    same(part, got[:, 5:16])


# ------------------------------------------------------------------ first-order reduction

def reduce_dev(ctx, blocks, n, flags=0, rank=0, world=1):
    from varsens_amd import _native
    nb, rows, l = blocks.shape
    k = (nb - 2) // 2
    f = ctx.upload_blocks(blocks)
    acc = ctx.alloc(8 * _native.acc_len(k, l, world))
    ctx.reduce_first_order(f, rows, k, l, acc, flags, rank, world)
    out = ctx.alloc(8 * _native.first_order_len(k, l))
    ctx.finish_first_order(acc, n, k, l, out, world)
    rec = ctx.download(acc, (_native.acc_len(k, l, world),))
    fin = ctx.download(out, (2 + 4 * k, l))
    return rec, fin


def unpack(rec, k, l):
    from varsens_amd import _native
    per = _native.acc_per_output(k)
    body = rec[4:4 + l * per].reshape(l, per)
    return dict(rows=rec[0], dropped=rec[1], sab=body[:, 0], sa=body[:, 1], sb=body[:, 2],
                ac=body[:, 3:3 + k].T, bd=body[:, 3 + k:3 + 2 * k].T, ad=body[:, 3 + 2 * k:3 + 3 * k].T,
                bc=body[:, 3 + 3 * k:3 + 4 * k].T, slots=rec[4 + l * per:])


def check_against_exact(rec, fin, blocks, n, rtol_acc=1e-13):
    nb, rows, l = blocks.shape
    k = (nb - 2) // 2
    ex = orc.exact_accumulators(blocks[0], blocks[1], blocks[2:2 + k], blocks[2 + k:])
    got = unpack(rec, k, l)
    assert got["rows"] == rows
    for name in ("sab", "sa", "sb", "ac", "bd", "ad", "bc"):
        scale = np.abs(ex[name]).max() + 1e-300
        np.testing.assert_allclose(got[name], ex[name], rtol=rtol_acc, atol=rtol_acc * scale, err_msg=name)
    want = orc.indices_from_accumulators(ex, n)
    np.testing.assert_allclose(fin[0], want["E_2"], rtol=1e-13)
    np.testing.assert_allclose(fin[1], want["var_y"], rtol=1e-12)
    np.testing.assert_allclose(fin[2:2 + k], want["U_j"], rtol=1e-13)
    np.testing.assert_allclose(fin[2 + k:2 + 2 * k], want["U_nj"], rtol=1e-13)
    return want


@pytest.mark.parametrize("k,rows,l", [(6, 1024, 1), (6, 1023, 1), (3, 1, 1), (3, 2, 1), (5, 7, 2), (7, 50, 3),
                                      (32, 4096, 1), (32, 20000, 1), (2, 70001, 1), (130, 600, 1), (4, 3000, 5)])
def test_reduce_random_values(ctx, k, rows, l):
    rng = np.random.RandomState(1000 * k + rows + l)
    blocks = 2.0 + rng.randn(2 * k + 2, rows, l)
    rec, fin = reduce_dev(ctx, blocks, max(rows, 2))
    check_against_exact(rec, fin, blocks, max(rows, 2))


def test_reduce_large_offset_variance(ctx):
    """var_y must survive a mean that dwarfs the spread (shifted sums, not raw moments)."""
    rng = np.random.RandomState(5)
    blocks = 1e6 + rng.randn(2 * 4 + 2, 30000, 1)
    rec, fin = reduce_dev(ctx, blocks, 30000)
    want = np.var(np.concatenate([blocks[0], blocks[1]]), axis=0, ddof=1)
    np.testing.assert_allclose(fin[1], want, rtol=1e-10)


def test_reduce_gfunction_matches_reference_golden(ctx, golden):
    """Config 1: device evaluation + device reduction against the reference's own numbers."""
    from varsens_amd import _native
    g = golden("g1_gfunc_k6_n1024")
    d = make_design(ctx, 6, 1024)
    f, blocks = eval_dev(ctx, d, _native.OBJ_G_FUNCTION, 6, 1, A6, 1024)
    rec, fin = reduce_dev(ctx, blocks, 1024)
    check_against_exact(rec, fin, blocks, 1024)
    k = 6
    # raw estimates: 5e-15 is the reference's own distance to correctly rounded sums at n=1024
    np.testing.assert_allclose(fin[0], g["E_2"], rtol=5e-15)
    np.testing.assert_allclose(fin[1], g["var_y"], rtol=5e-15)
    np.testing.assert_allclose(fin[2:2 + k], g["U_j"], rtol=5e-15)
    np.testing.assert_allclose(fin[2 + k:2 + 2 * k], g["U_nj"], rtol=5e-15)
    # indices: north-star tolerance 1e-10 relative
    np.testing.assert_allclose(fin[2 + 2 * k:2 + 3 * k], g["sens"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(fin[2 + 3 * k:2 + 4 * k], g["sens_t"], rtol=1e-10, atol=1e-14)


def test_reduce_nan_trim_golden(ctx, golden):
    from varsens_amd import _native
    g = golden("g6_nantrim_k5_n64_l2")
    k, n, l = 5, 64, 2
    blocks = g["vals"].reshape(2 * k + 2, n, l)
    rec, fin = reduce_dev(ctx, blocks, n, flags=_native.REDUCE_NAN_TRIM)
    assert rec[0] == 61 and rec[1] == 3
    # output 0 is finite and equals the reference (divisor stays n = 64); output 1 keeps its NaNs
    np.testing.assert_allclose(fin[0][0], g["E_2"][0], rtol=1e-14)
    np.testing.assert_allclose(fin[1][0], g["var_y"][0], rtol=1e-13)
    np.testing.assert_allclose(fin[2 + 2 * k:2 + 3 * k, 0], g["sens"][:, 0], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(fin[2 + 3 * k:2 + 4 * k, 0], g["sens_t"][:, 0], rtol=1e-10, atol=1e-14)
    assert np.isnan(fin[2 + 2 * k:2 + 3 * k, 1]).all()
    # without the flag the NaNs poison everything, as on the reference's live-evaluation route
    rec2, fin2 = reduce_dev(ctx, blocks, n)
    assert np.isnan(fin2[0]).all()


def test_reduce_nan_trim_big_fast_path(ctx):
    from varsens_amd import _native
    rng = np.random.RandomState(77)
    k, rows = 9, 20000
    blocks = 1.0 + rng.rand(2 * k + 2, rows, 1)
    bad = rng.choice(rows, 37, replace=False)
    blocks[rng.randint(0, 2 * k + 2, 37), bad, 0] = np.nan
    blocks[0, 0, 0] = np.nan   # the row the un-masked pass takes its shift from
    rec, fin = reduce_dev(ctx, blocks, rows, flags=_native.REDUCE_NAN_TRIM)
    keep = ~np.isnan(blocks[:, :, 0]).any(axis=0)
    kept = blocks[:, keep]
    assert rec[0] == keep.sum() and rec[1] == rows - keep.sum()
    ex = orc.exact_accumulators(kept[0], kept[1], kept[2:2 + k], kept[2 + k:])
    got = unpack(rec, k, 1)
    for name in ("sab", "ac", "bd", "ad", "bc"):
        np.testing.assert_allclose(got[name], ex[name], rtol=1e-13)
    want = orc.indices_from_accumulators(ex, rows)
    np.testing.assert_allclose(fin[1], want["var_y"], rtol=1e-12)


def test_reduce_is_bit_reproducible(ctx):
    rng = np.random.RandomState(3)
    blocks = rng.randn(2 * 16 + 2, 50000, 1)
    r1, _ = reduce_dev(ctx, blocks, 50000)
    r2, _ = reduce_dev(ctx, blocks, 50000)
    same(r1, r2)


def test_reduce_shards_merge_like_one_pass(ctx):
    """Two ranks' records summed (what the all-reduce does) finish to the single-GPU result."""
    from varsens_amd import _native
    rng = np.random.RandomState(11)
    k, rows, l = 6, 9000, 2
    blocks = 5.0 + rng.randn(2 * k + 2, rows, l)
    blocks[:, :4000] += 3.0   # make the shard means differ
    rec, fin = reduce_dev(ctx, blocks, rows)
    world = 2
    recs = []
    for rank, (b, e) in enumerate([(0, 4000), (4000, rows)]):
        f = ctx.upload_blocks(blocks[:, b:e])
        acc = ctx.alloc(8 * _native.acc_len(k, l, world))
        ctx.reduce_first_order(f, e - b, k, l, acc, 0, rank, world)
        recs.append(ctx.download(acc, (_native.acc_len(k, l, world),)))
    summed = ctx.upload(recs[0] + recs[1])
    out = ctx.alloc(8 * _native.first_order_len(k, l))
    ctx.finish_first_order(summed, rows, k, l, out, world)
    fin2 = ctx.download(out, (2 + 4 * k, l))
    np.testing.assert_allclose(fin2, fin, rtol=1e-12)
    np.testing.assert_allclose(fin2[1], np.var(np.concatenate([blocks[0], blocks[1]]), axis=0, ddof=1), rtol=1e-12)


def test_rccl_single_rank_allreduce(ctx):
    """dlopen(librccl) + ncclCommInitRank + ncclAllReduce on the engine's stream (1 rank: identity)."""
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 1, 0)
    x = np.arange(137, dtype=float)
    buf = ctx.upload(x)
    ctx.comm_allreduce(buf, 137)
    ctx.sync()
    same(ctx.download(buf, (137,)), x)
    ctx.comm_destroy()


# ------------------------------------------------------------------ second order (fp64 MFMA Gram)

@pytest.mark.parametrize("k,rows,l", [(6, 1000, 1), (32, 5000, 1), (40, 3001, 1), (70, 777, 1), (7, 500, 3),
                                      (3, 5, 1), (33, 64, 2), (6, 200000, 1), (12, 100003, 1), (24, 65536, 1),
                                      (64, 4099, 1), (65, 130, 1), (128, 2050, 1), (200, 97, 1), (50, 1000, 3), (129, 15, 1),
                                      # 32-sample-row staging (one output, even row count): ragged ends of every kind
                                      (17, 34, 1), (31, 30, 1), (28, 2, 1), (33, 46, 1), (64, 48, 1), (96, 62, 1),
                                      (100, 4128, 1), (32, 65598, 1),
                                      # several outputs on the two-chunk S/T kernel (33..64 index rows per half, even rows)
                                      (20, 1000, 3), (11, 4096, 4), (16, 62, 4), (13, 3000, 3),
                                      # gram_st_kernel (round 3): every count of live 16-row blocks, partly filled last
                                      # blocks, one and many stages per workgroup, ragged last stage
                                      (50, 200, 1), (56, 1000, 1), (63, 70, 1), (72, 2048, 1), (90, 330, 1), (112, 64, 1),
                                      (120, 500, 1), (127, 34, 1), (64, 100002, 1), (128, 70000, 1), (80, 40000, 1),
                                      # ... with several outputs (index row = output * k + column; 65..128 rows per half)
                                      (32, 2048, 4), (24, 1000, 5), (22, 70, 3), (43, 500, 2), (9, 40000, 9), (16, 62, 8),
                                      (37, 100002, 3), (1, 3000, 128), (2, 64, 33)])
def test_second_order_gram(ctx, k, rows, l):
    rng = np.random.RandomState(k * 7 + rows + l)
    blocks = 1.0 + rng.randn(2 * k + 2, rows, l)
    f = ctx.upload_blocks(blocks)
    K = k * l
    gram = ctx.alloc(8 * 2 * K * K)
    ctx.reduce_second_order(f, rows, k, l, gram)
    G = ctx.download(gram, (2, K, K))
    C = blocks[2:2 + k].transpose(0, 2, 1).reshape(K, rows)
    D = blocks[2 + k:].transpose(0, 2, 1).reshape(K, rows)
    S, T = C + D, C - D          # the interface's sums: SS = S S^T, TT = T T^T (include/varsens_hip.h)
    scale = np.abs(S @ S.T).max()
    np.testing.assert_allclose(G[0], S @ S.T, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(G[1], T @ T.T, rtol=1e-12, atol=1e-12 * scale)
    assert np.array_equal(G[0], G[0].T) and np.array_equal(G[1], G[1].T)
    # what the reference forms from its four tensordots (saltelli.py:612-619)
    np.testing.assert_allclose(0.5 * (G[0] - G[1]), D @ C.T + C @ D.T, rtol=1e-12, atol=2e-12 * scale)
    np.testing.assert_allclose(0.5 * (G[0] + G[1]), D @ D.T + C @ C.T, rtol=1e-12, atol=2e-12 * scale)
    # exactness spot check against correctly rounded sums
    import math
    for (i, j) in ((0, 0), (K - 1, 0), (K // 2, K - 1)):
        want = math.fsum((S[i] * S[j]).tolist())
        assert abs(G[0][i, j] - want) <= 1e-13 * max(1.0, abs(want)) * 10


def _ctx_with_gram_dma(value):
    import os
    from varsens_amd import _native
    old = {v: os.environ.get(v) for v in ("VS_TUNING", "VS_GRAM_DMA")}
    os.environ.update(VS_TUNING="1", VS_GRAM_DMA=value)
    try:
        return _native.Context(0)
    finally:
        for v, x in old.items():
            if x is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = x


@pytest.fixture(scope="module")
def ctx_dma_all():
    """gram_dma_kernel also for 17 .. 32 index rows (VS_GRAM_DMA=2; by default it starts at 33)"""
    c = _ctx_with_gram_dma("2")
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_no_dma():
    """A context that keeps 17 .. 64 index rows per half on gram_diag_wide_kernel / gram_st_kernel (VS_GRAM_DMA=0): the
    kernels gram_dma_kernel replaced for second order alone."""
    c = _ctx_with_gram_dma("0")
    yield c
    c.close()


# gram_dma_kernel (csrc/vs_gram_dma.hip): every count of live blocks and of live rows in the last piece of eight, no
# whole unit at all, a partial last unit of every even length, many units per wave, a short last row block
@pytest.mark.parametrize("k,rows", [(33, 2), (33, 16), (33, 18), (34, 4110), (39, 32), (40, 14), (40, 4096), (41, 2050),
                                    (47, 300), (48, 32), (48, 5000), (49, 30), (50, 2062), (55, 1024), (57, 2064),
                                    (62, 40), (63, 66), (64, 2), (64, 16), (64, 30000), (64, 131078), (36, 262144),
                                    (56, 2048 * 5 + 20),
                                    # 17 .. 32 index rows: sixteen waves per CU on images of 32 rows per half
                                    (17, 2), (17, 34), (20, 16), (23, 4100), (24, 32), (24, 70000), (25, 30), (31, 2062),
                                    (32, 2), (32, 16), (32, 48), (32, 65598), (32, 262144), (29, 2048 * 3 + 8)])
def test_second_order_dma_kernel(ctx_dma_all, ctx_no_dma, k, rows):
    """17 .. 64 index rows per half, one output, even row count: S S^T and T T^T from gram_dma_kernel against numpy,
    against exactly rounded sums, against the kernel it replaced, and bit for bit run to run."""
    import math
    rng = np.random.RandomState(k * 31 + rows)
    blocks = 1.0 + rng.randn(2 * k + 2, rows, 1)
    C, D = blocks[2:2 + k, :, 0], blocks[2 + k:, :, 0]
    S, T = C + D, C - D
    got = []
    for c in (ctx_dma_all, ctx_dma_all, ctx_no_dma):
        f = c.upload_blocks(blocks)
        gram = c.alloc(8 * 2 * k * k)
        c.reduce_second_order(f, rows, k, 1, gram)
        got.append(c.download(gram, (2, k, k)))
        f.free()
        gram.free()
    G = got[0]
    assert np.array_equal(G.view(np.uint64), got[1].view(np.uint64)), "not the same bits run to run"
    scale = np.abs(S @ S.T).max()
    np.testing.assert_allclose(G[0], S @ S.T, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(G[1], T @ T.T, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(G, got[2], rtol=1e-12, atol=1e-12 * scale)
    assert np.array_equal(G[0], G[0].T) and np.array_equal(G[1], G[1].T)
    for (i, j) in ((0, 0), (k - 1, 0), (k // 2, k - 1), (k - 1, k - 1), (15, 16), (31, 32)):
        if i >= k or j >= k:
            continue
        for M, X in ((G[0], S), (G[1], T)):
            want = math.fsum((X[i] * X[j]).tolist())
            assert abs(M[i, j] - want) <= 1e-13 * scale, (i, j)
    # both orders in one call: from 33 index rows on the same kernel carries the first-order dot products (lane u owns
    # index row u); the record against the first-order kernel's and exact sums
    from varsens_amd import _native
    nacc = _native.acc_len(k, 1, 1)
    f = ctx_dma_all.upload_blocks(blocks)
    acc1, acc2, g2 = ctx_dma_all.alloc(8 * nacc), ctx_dma_all.alloc(8 * nacc), ctx_dma_all.alloc(8 * 2 * k * k)
    ctx_dma_all.reduce_first_order(f, rows, k, 1, acc1)
    ctx_dma_all.reduce_first_second_order(f, rows, k, 1, acc2, g2)
    a1, a2 = ctx_dma_all.download(acc1, (nacc,)), ctx_dma_all.download(acc2, (nacc,))
    G2 = ctx_dma_all.download(g2, (2, k, k))
    for b in (f, acc1, acc2, g2):
        b.free()
    assert np.array_equal(a1[:4], a2[:4])
    per = _native.acc_per_output(k)
    r1, r2 = a1[_native.ACC_HEADER:_native.ACC_HEADER + per], a2[_native.ACC_HEADER:_native.ACC_HEADER + per]
    np.testing.assert_allclose(r2, r1, rtol=2e-13, atol=1e-13 * np.abs(r1).max())
    a, b = blocks[0, :, 0], blocks[1, :, 0]
    for j in (0, k // 2, k - 1):
        for got, want in ((r2[3 + j], math.fsum((a * C[j]).tolist())), (r2[3 + k + j], math.fsum((b * D[j]).tolist())),
                          (r2[3 + 2 * k + j], math.fsum((a * D[j]).tolist())), (r2[3 + 3 * k + j], math.fsum((b * C[j]).tolist()))):
            assert abs(got - want) <= 1e-13 * max(abs(want), np.abs(r1).max() * 1e-3), j
    np.testing.assert_allclose(G2, G, rtol=1e-12, atol=1e-12 * scale)


# ... with several outputs (index row u = column * l + output lives in plane output * (2k + 2) + 2 + column): 33 .. 64 rows
@pytest.mark.parametrize("k,l,rows", [(17, 2, 34), (20, 3, 1000), (11, 4, 4096), (16, 4, 62), (13, 3, 3000), (32, 2, 65598),
                                      (9, 7, 2050), (21, 3, 4110), (8, 8, 30), (12, 5, 131078), (16, 3, 48), (7, 9, 100),
                                      (2, 17, 66), (1, 64, 2064), (1, 33, 16), (5, 11, 2), (19, 2, 2048 * 3 + 6)])
def test_second_order_dma_kernel_several_outputs(ctx, ctx_no_dma, k, l, rows):
    rng = np.random.RandomState(k * 131 + l * 17 + rows)
    blocks = 1.0 + rng.randn(2 * k + 2, rows, l)
    K = k * l
    C = blocks[2:2 + k].transpose(0, 2, 1).reshape(K, rows)
    D = blocks[2 + k:].transpose(0, 2, 1).reshape(K, rows)
    S, T = C + D, C - D
    got = []
    for c in (ctx, ctx, ctx_no_dma):
        f = c.upload_blocks(blocks)
        gram = c.alloc(8 * 2 * K * K)
        c.reduce_second_order(f, rows, k, l, gram)
        got.append(c.download(gram, (2, K, K)))
        f.free()
        gram.free()
    G = got[0]
    assert np.array_equal(G.view(np.uint64), got[1].view(np.uint64)), "not the same bits run to run"
    scale = np.abs(S @ S.T).max()
    np.testing.assert_allclose(G[0], S @ S.T, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(G[1], T @ T.T, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(G, got[2], rtol=1e-12, atol=1e-12 * scale)
    assert np.array_equal(G[0], G[0].T) and np.array_equal(G[1], G[1].T)


@pytest.mark.parametrize("k,rows,l", [(6, 1000, 1), (32, 5000, 1), (16, 200000, 1), (3, 5, 1), (12, 100003, 1),
                                      (1, 17, 1), (25, 4096, 1), (40, 3000, 1), (5, 700, 2), (33, 2, 1), (64, 4099, 1),
                                      (65, 777, 1), (128, 10000, 1), (129, 2050, 1), (200, 333, 1), (48, 65536, 1),
                                      (17, 34, 1), (31, 30, 1), (32, 65598, 1), (33, 46, 1), (64, 48, 1), (96, 62, 1),
                                      # gram_st_kernel carrying the first-order dot products (round 3)
                                      (50, 200, 1), (56, 1000, 1), (63, 70, 1), (72, 2048, 1), (90, 330, 1), (112, 64, 1),
                                      (120, 500, 1), (127, 34, 1), (64, 100002, 1), (128, 70000, 1), (80, 40000, 1)])
def test_first_and_second_order_in_one_pass(ctx, k, rows, l):
    """vs_reduce_first_second_order fills the record and the Gram matrices of the two separate passes
    (l == 1: k <= 32 the wave-private kernel, beyond that the macro kernel's diagonal pairs carry the dot
    products and a small kernel the scalar sums; other shapes: the separate kernels behind the same call)."""
    from varsens_amd import _native
    rng = np.random.RandomState(k * 13 + rows + l)
    blocks = 3.0 + rng.randn(2 * k + 2, rows, l)
    f = ctx.upload_blocks(blocks)
    K = k * l
    nacc = _native.acc_len(k, l, 1)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    g1, g2 = ctx.alloc(8 * 2 * K * K), ctx.alloc(8 * 2 * K * K)
    ctx.reduce_first_order(f, rows, k, l, acc1)
    ctx.reduce_second_order(f, rows, k, l, g1)
    ctx.reduce_first_second_order(f, rows, k, l, acc2, g2)
    a1, a2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    G1, G2 = ctx.download(g1, (2, K, K)), ctx.download(g2, (2, K, K))
    assert np.array_equal(a1[:4], a2[:4])
    np.testing.assert_allclose(a2, a1, rtol=2e-13, atol=1e-9)
    np.testing.assert_allclose(G2, G1, rtol=1e-13, atol=1e-13 * np.abs(G1).max())
    first = ctx.alloc(8 * _native.first_order_len(k, l))
    first1 = ctx.alloc(8 * _native.first_order_len(k, l))
    ctx.finish_first_order(acc2, rows, k, l, first)
    ctx.finish_first_order(acc1, rows, k, l, first1)
    np.testing.assert_allclose(ctx.download(first, (2 + 4 * k, l)), ctx.download(first1, (2 + 4 * k, l)),
                               rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("k,rows", [(20, 5000), (32, 100000), (40, 3000), (128, 2050)])
def test_wide_staging_agrees_with_the_chunk_kernels(ctx, k, rows, monkeypatch):
    """VS_GRAM_WIDE=0 (read once per context) keeps the 16-sample-row staging for every shape: same sums."""
    from varsens_amd import _native
    rng = np.random.RandomState(k + rows)
    blocks = 2.0 + rng.randn(2 * k + 2, rows, 1)
    nacc = _native.acc_len(k, 1, 1)
    narrow = tuned_context(monkeypatch, VS_GRAM_WIDE=0)
    out = []
    for c in (ctx, narrow):
        f = c.upload_blocks(blocks)
        acc, g1, g2 = c.alloc(8 * nacc), c.alloc(8 * 2 * k * k), c.alloc(8 * 2 * k * k)
        c.reduce_second_order(f, rows, k, 1, g1)
        c.reduce_first_second_order(f, rows, k, 1, acc, g2)
        out.append((c.download(g1, (2, k, k)), c.download(g2, (2, k, k)), c.download(acc, (nacc,))))
        for b in (f, acc, g1, g2):
            b.free()
    narrow.close()
    scale = np.abs(out[1][0]).max()   # TT = (C - D)(C - D)^T holds sums of zero-mean products: absolute bound
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-13, atol=1e-13 * scale)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-13, atol=1e-13 * scale)
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=2e-13, atol=1e-9)


def test_one_pass_with_nan_trim_takes_the_separate_kernels(ctx):
    from varsens_amd import _native
    rng = np.random.RandomState(3)
    k, rows = 6, 1200
    blocks = 1.0 + rng.rand(2 * k + 2, rows, 1)
    blocks[3, 17, 0] = np.nan
    blocks[0, 900, 0] = np.nan
    f = ctx.upload_blocks(blocks)
    nacc = _native.acc_len(k, 1, 1)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    g1, g2 = ctx.alloc(8 * 2 * k * k), ctx.alloc(8 * 2 * k * k)
    ctx.reduce_first_order(f, rows, k, 1, acc1, flags=_native.REDUCE_NAN_TRIM)
    ctx.reduce_second_order(f, rows, k, 1, g1, flags=_native.REDUCE_NAN_TRIM)
    ctx.reduce_first_second_order(f, rows, k, 1, acc2, g2, flags=_native.REDUCE_NAN_TRIM)
    assert np.array_equal(ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,)))
    assert np.array_equal(ctx.download(g1, (2, k, k)), ctx.download(g2, (2, k, k)))
    assert ctx.download(acc2, (nacc,))[1] == 2.0     # rows dropped
    # without the flag the combined kernel runs and the NaNs propagate, as in the separate passes
    ctx.reduce_first_second_order(f, rows, k, 1, acc2, g2)
    assert np.isnan(ctx.download(acc2, (nacc,))).any() and np.isnan(ctx.download(g2, (2, k, k))).any()


@pytest.mark.parametrize("k,rows", [(5, 900), (9, 50001), (40, 3001), (70, 1000), (129, 257)])   # k > 32: macro-tile kernel
def test_second_order_nan_trim(ctx, k, rows):
    from varsens_amd import _native
    rng = np.random.RandomState(9)
    blocks = 1.0 + rng.rand(2 * k + 2, rows, 1)
    bad = rng.choice(rows, 11, replace=False)
    blocks[rng.randint(0, 2 * k + 2, 11), bad, 0] = np.nan
    f = ctx.upload_blocks(blocks)
    gram = ctx.alloc(8 * 2 * k * k)
    ctx.reduce_second_order(f, rows, k, 1, gram, flags=_native.REDUCE_NAN_TRIM)
    G = ctx.download(gram, (2, k, k))
    keep = ~np.isnan(blocks[:, :, 0]).any(axis=0)
    C = blocks[2:2 + k, keep, 0]; D = blocks[2 + k:, keep, 0]
    np.testing.assert_allclose(G[0], (C + D) @ (C + D).T, rtol=1e-12)
    np.testing.assert_allclose(G[1], (C - D) @ (C - D).T, rtol=1e-12, atol=1e-12 * np.abs(G[0]).max())
    ctx.reduce_second_order(f, rows, k, 1, gram)      # without the flag NaNs propagate
    assert np.isnan(ctx.download(gram, (2, k, k))).any()


@pytest.mark.parametrize("block", [0, 1])
@pytest.mark.parametrize("k,rows,l", [(5, 900, 1), (20, 4097, 1), (3, 700, 2), (40, 1001, 1), (17, 600, 3)])
def test_second_order_nan_only_in_fM(ctx, block, k, rows, l):
    """The lone NaN sits in fM_1 or fM_2, blocks the Gram matrix X = [C; D] never reads: the row is
    still dropped from fN_j / fN_nj (varsens/saltelli.py:474-492 trims all four arrays)."""
    from varsens_amd import _native
    rng = np.random.RandomState(31 + block)
    blocks = 1.0 + rng.rand(2 * k + 2, rows, l)
    r_bad = rows // 3
    blocks[block, r_bad, 0] = np.nan
    f = ctx.upload_blocks(blocks)
    K = k * l
    gram = ctx.alloc(8 * 2 * K * K)
    ctx.reduce_second_order(f, rows, k, l, gram, flags=_native.REDUCE_NAN_TRIM)
    G = ctx.download(gram, (2, K, K))
    keep = np.ones(rows, bool); keep[r_bad] = False
    C = blocks[2:2 + k][:, keep].transpose(0, 2, 1).reshape(K, -1)      # index row j*l + o
    D = blocks[2 + k:][:, keep].transpose(0, 2, 1).reshape(K, -1)
    np.testing.assert_allclose(G[0], (C + D) @ (C + D).T, rtol=1e-12)
    np.testing.assert_allclose(G[1], (C - D) @ (C - D).T, rtol=1e-12, atol=1e-12 * np.abs(G[0]).max())
    # the untrimmed sums differ: the test would not notice a skipped trim otherwise
    Cu = blocks[2:2 + k].transpose(0, 2, 1).reshape(K, -1)
    Du = blocks[2 + k:].transpose(0, 2, 1).reshape(K, -1)
    assert np.max(np.abs(G[0] - (Cu + Du) @ (Cu + Du).T)) > 0.5
    # both orders through the one call Varsens uses
    nacc = _native.acc_len(k, l, 1)
    acc, g2 = ctx.alloc(8 * nacc), ctx.alloc(8 * 2 * K * K)
    ctx.reduce_first_second_order(f, rows, k, l, acc, g2, flags=_native.REDUCE_NAN_TRIM)
    assert np.array_equal(ctx.download(g2, (2, K, K)), G)
    assert ctx.download(acc, (nacc,))[1] == 1.0


def test_two_stream_pipeline_matches_serial(ctx):
    """Exchange/finish of pass i on the auxiliary stream beside pass i+1 on the main stream (bench.py's
    schedule) gives the results of the serial schedule."""
    from varsens_amd import _native
    rng = np.random.RandomState(21)
    k, rows = 8, 40000
    data = [1.0 + rng.rand(2 * k + 2, rows, 1) for _ in range(4)]
    serial = [reduce_dev(ctx, d, rows)[1] for d in data]
    bufs = [ctx.upload_blocks(d) for d in data]
    accs = [ctx.alloc(8 * _native.acc_len(k, 1, 1)) for _ in range(2)]
    outs = [ctx.alloc(8 * _native.first_order_len(k, 1)) for _ in range(4)]
    for i, f in enumerate(bufs):
        b = i & 1
        ctx.stream_select(0)
        ctx.stream_wait_event(0, b)
        ctx.reduce_first_order(f, rows, k, 1, accs[b])
        ctx.event_record(2 + b, 0)
        ctx.stream_wait_event(1, 2 + b)
        ctx.stream_select(1)
        ctx.finish_first_order(accs[b], rows, k, 1, outs[i])
        ctx.event_record(b, 1)
        ctx.stream_select(0)
    ctx.sync()
    for i in range(4):
        same(ctx.download(outs[i], (2 + 4 * k, 1)), serial[i])


# ------------------------------------------------------------------ fused evaluate + reduce

@pytest.mark.parametrize("kind,k,n", [("g", 6, 5000), ("g", 32, 20001), ("lin", 32, 70000), ("poly", 16, 333),
                                      ("lin", 1, 64), ("g", 31, 1)])
def test_fused_matches_two_kernel_path(ctx, kind, k, n):
    from varsens_amd import _native
    rng = np.random.RandomState(k + n)
    params = {"g": rng.rand(k) * 5, "lin": np.arange(1, k + 1, dtype=float), "poly": 1.0 / (1.0 + np.arange(k) / 8.0)}[kind]
    code = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[kind]
    d = make_design(ctx, k, n)
    obj = ctx.objective_create(code, k, 1, params)
    nacc = _native.acc_len(k, 1, 1)
    f = ctx.alloc(8 * 2 * (k + 1) * n)
    ctx.eval(d, obj, 0, n, f)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    ctx.reduce_first_order(f, n, k, 1, acc1)
    ctx.eval_reduce_first_order(d, obj, 0, n, acc2)
    r1, r2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    assert r1[0] == r2[0] == n
    np.testing.assert_allclose(r2[4:4 + 3 + 4 * k], r1[4:4 + 3 + 4 * k], rtol=1e-13)
    if n > 1:
        out1, out2 = ctx.alloc(8 * (2 + 4 * k)), ctx.alloc(8 * (2 + 4 * k))
        ctx.finish_first_order(acc1, n, k, 1, out1)
        ctx.finish_first_order(acc2, n, k, 1, out2)
        a, b = ctx.download(out1, (2 + 4 * k,)), ctx.download(out2, (2 + 4 * k,))
        np.testing.assert_allclose(b[:2 + 2 * k], a[:2 + 2 * k], rtol=1e-12)
        np.testing.assert_allclose(b[2 + 2 * k:], a[2 + 2 * k:], rtol=1e-9, atol=1e-12)


def test_fused_refuses_what_it_does_not_cover(ctx):
    from varsens_amd import _native
    d = make_design(ctx, 200, 100)
    obj = ctx.objective_create(_native.OBJ_LINEAR, 200, 1, np.ones(200))       # beyond the fused kernels' 128 columns
    acc = ctx.alloc(8 * _native.acc_len(200, 1, 1))
    assert not ctx.eval_reduce_supported(d, obj)
    with pytest.raises(_native.VarsensHipError, match="fused path covers"):
        ctx.eval_reduce_first_order(d, obj, 0, 100, acc)
    d8 = make_design(ctx, 8, 100)
    inv = ctx.objective_create(_native.OBJ_INVERT, 8, 8, None)                 # not separable
    assert not ctx.eval_reduce_supported(d8, inv)
    lo, hi = np.full(8, 1.0), np.full(8, 10.0)
    dp = make_design(ctx, 8, 100, _native.SCALE_POWER, lo, hi)                 # power scaling: built-ins have fused instances
    lin = ctx.objective_create(_native.OBJ_LINEAR, 8, 1, np.ones(8))
    assert ctx.eval_reduce_supported(dp, lin) and ctx.eval_reduce_supported(d8, lin)


@pytest.mark.parametrize("kind,k,n,l", [("g", 6, 5000, 1), ("lin", 32, 3001, 1), ("poly", 70, 1000, 2), ("g", 128, 700, 1)])
def test_fused_power_scaling_matches_two_kernel_path(ctx, kind, k, n, l):
    """scale.Power designs through the fused evaluate+reduce kernels (round 3: own instances, GENV 3): the scaled
    coordinates come from the same out-of-line pow() as vs_eval's, so the record differs from the two-kernel
    path's by summation order only."""
    from varsens_amd import _native
    rng = np.random.RandomState(k + n)
    one = {"g": rng.rand(k) * 5, "lin": np.arange(1, k + 1, dtype=float), "poly": 1.0 / (1.0 + np.arange(k) / 8.0)}[kind]
    params = np.stack([one, one[::-1]])[:l]
    code = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[kind]
    lo, hi = np.full(k, 1e-2), np.linspace(0.5, 1.0, k)       # scale.py:62: lower * (upper / lower) ** x
    d = make_design(ctx, k, n, _native.SCALE_POWER, lo, hi)
    obj = ctx.objective_create(code, k, l, params)
    assert ctx.eval_reduce_supported(d, obj)
    nacc = _native.acc_len(k, l, 1)
    f = ctx.alloc(8 * 2 * (k + 1) * n * l)
    ctx.eval(d, obj, 0, n, f)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    ctx.reduce_first_order(f, n, k, l, acc1)
    ctx.eval_reduce_first_order(d, obj, 0, n, acc2)
    r1, r2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    per = _native.acc_per_output(k)
    np.testing.assert_allclose(r2[4:4 + l * per], r1[4:4 + l * per], rtol=2e-13)
    # and against the oracle's power scaling (numpy pow vs device pow: a few ulp per coordinate)
    S = orc.build_sample(k, n, lambda x: orc.scale_power(x, lo, hi))
    name = {"g": "g_function", "lin": "linear", "poly": "polynomial"}[kind]
    ex = orc.exact_accumulators(*orc.evaluate_engine_order(name, params, S[0], S[1]))
    np.testing.assert_allclose(r2[4 + 3:4 + 3 + k], ex["ac"][:, 0], rtol=1e-12)


@pytest.mark.parametrize("kind,k,n,mode", [
    ("g", 6, 1024, 0), ("lin", 32, 4096, 0), ("poly", 32, 70001, 1), ("g", 17, 333, 0), ("lin", 16, 5000, 1),
    ("poly", 1, 64, 0), ("g", 3, 1, 0), ("lin", 9, 63, 0), ("g", 24, 65, 2), ("poly", 31, 20000, 0), ("lin", 12, 129, 2),
    ("g", 32, 1 << 18, 0), ("lin", 2, 100003, 0), ("g", 8, 300000, 1), ("poly", 20, 640, 0),
    # 113 .. 128 columns: the eight-wave kernel (137 KiB tile, nine accumulator tiles per wave)
    ("poly", 128, 5000, 0), ("lin", 113, 700, 1), ("g", 120, 130, 0), ("lin", 125, 20000, 2), ("g", 128, 1, 0),
    ("poly", 114, 64, 0), ("lin", 119, 63, 0), ("g", 127, 40000, 0), ("poly", 128, 1 << 17, 1),
    # 49 .. 64 columns: four waves x 16 columns (69 KiB tile, five accumulator tiles per wave)
    ("lin", 64, 5000, 0), ("g", 49, 700, 1), ("poly", 52, 130, 0), ("lin", 60, 20000, 2), ("g", 64, 1, 0),
    ("poly", 50, 64, 0), ("lin", 57, 63, 0), ("g", 63, 40000, 0), ("poly", 64, 1 << 17, 1),
    # dead 16-row blocks in the tile (bodies compiled per count of live blocks): 3 of 4, 5 / 6 / 7 of 8
    ("lin", 33, 700, 1), ("g", 40, 5000, 0), ("poly", 48, 130, 2), ("lin", 65, 3000, 0), ("g", 80, 64, 1),
    ("poly", 81, 20000, 0), ("lin", 96, 4097, 2), ("g", 97, 1, 0), ("poly", 112, 1000, 0), ("lin", 100, 1 << 16, 0)])
def test_fused_both_orders_matches_two_kernel_path(ctx, kind, k, n, mode):
    """vs_eval_reduce_first_second_order (round 3: s/t rows of every 64-row block into an LDS tile, MFMA Gram and the
    first-order dot products in the evaluating workgroup) == vs_eval + vs_reduce_first_second_order up to summation
    order, for every scaling mode, ragged last blocks, one workgroup and many, k on both sides of 16."""
    from varsens_amd import _native
    rng = np.random.RandomState(k * 31 + n)
    params = {"g": rng.rand(k) * 5, "lin": np.arange(1, k + 1, dtype=float), "poly": 1.0 / (1.0 + np.arange(k) / 8.0)}[kind]
    code = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[kind]
    lo = hi = None
    if mode == 1:
        lo, hi = np.linspace(-1, 0.5, k), np.linspace(1, 3, k)
    elif mode == 2:
        lo, hi = np.full(k, 1e-2), np.linspace(0.5, 1.0, k)
    d = make_design(ctx, k, n, mode, lo, hi)
    obj = ctx.objective_create(code, k, 1, params)
    assert ctx.eval_reduce_both_supported(d, obj)
    nacc, ng = _native.acc_len(k, 1, 1), _native.gram_len(k)
    f = ctx.alloc(8 * 2 * (k + 1) * n)
    ctx.eval(d, obj, 0, n, f)
    acc1, acc2, g1, g2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc), ctx.alloc(8 * ng), ctx.alloc(8 * ng)
    ctx.reduce_first_second_order(f, n, k, 1, acc1, g1)
    ctx.eval_reduce_first_second_order(d, obj, 0, n, acc2, g2)
    r1, r2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    G1, G2 = ctx.download(g1, (2, k, k)), ctx.download(g2, (2, k, k))
    assert r2[0] == r1[0] == n and r2[1] == 0
    per = _native.acc_per_output(k)
    np.testing.assert_allclose(r2[4:4 + per], r1[4:4 + per], rtol=1e-12)
    np.testing.assert_allclose(r2[4 + per:], r1[4 + per:], rtol=1e-9, atol=1e-9 * abs(r1[4 + per + 1]))   # count, mean, m2
    scale = np.abs(G1).max()
    np.testing.assert_allclose(G2, G1, rtol=1e-12, atol=1e-13 * scale)
    assert np.array_equal(G2[0], G2[0].T) and np.array_equal(G2[1], G2[1].T)
    # and straight against the f-values: S S^T, T T^T in numpy
    F = ctx.download_blocks(f, k, 1, n)[:, :, 0]
    S, T = F[2:2 + k] + F[2 + k:], F[2:2 + k] - F[2 + k:]
    np.testing.assert_allclose(G2[0], S @ S.T, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(G2[1], T @ T.T, rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(r2[4 + 3:4 + 3 + k], F[2:2 + k] @ F[0], rtol=1e-12)            # a . c_i


@pytest.mark.parametrize("n", [1, 200, 4096, 100003, 1 << 20])
def test_fused_both_orders_ishigami(ctx, n):
    """BASELINE configs[2] (Ishigami, k = 3) through vs_eval_reduce_first_second_order: the thread-per-row fused kernel
    with the 12 second-order products beside its 17 first-order sums."""
    from varsens_amd import _native
    k = 3
    d = make_design(ctx, k, n, 1, np.full(3, -np.pi), np.full(3, np.pi))
    obj = ctx.objective_create(_native.OBJ_ISHIGAMI, 3, 1, np.array([7.0, 0.1]))
    assert ctx.eval_reduce_both_supported(d, obj)
    nacc, ng = _native.acc_len(k, 1, 1), _native.gram_len(k)
    f = ctx.alloc(8 * 8 * n)
    ctx.eval(d, obj, 0, n, f)
    acc1, acc2, g1, g2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc), ctx.alloc(8 * ng), ctx.alloc(8 * ng)
    ctx.reduce_first_second_order(f, n, k, 1, acc1, g1)
    ctx.eval_reduce_first_second_order(d, obj, 0, n, acc2, g2)
    r1, r2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    G1, G2 = ctx.download(g1, (2, k, k)), ctx.download(g2, (2, k, k))
    per = _native.acc_per_output(k)
    np.testing.assert_allclose(r2[4:4 + per], r1[4:4 + per], rtol=1e-12, atol=1e-12 * np.abs(r1[4:4 + per]).max())
    np.testing.assert_allclose(G2, G1, rtol=1e-12, atol=1e-12 * np.abs(G1).max())
    assert np.array_equal(G2[0], G2[0].T) and np.array_equal(G2[1], G2[1].T)
    F = ctx.download_blocks(f, 3, 1, n)[:, :, 0]
    S, T = F[2:5] + F[5:8], F[2:5] - F[5:8]
    np.testing.assert_allclose(G2[0], S @ S.T, rtol=1e-12, atol=1e-12 * np.abs(G1).max())
    np.testing.assert_allclose(G2[1], T @ T.T, rtol=1e-12, atol=1e-12 * np.abs(G1).max())


@pytest.mark.parametrize("k,n", [(6, 100003), (32, 1 << 16), (64, 50000), (128, 20000), (3, 70001)])
def test_fused_both_orders_is_bit_reproducible(ctx, k, n):
    """Fixed block-to-workgroup map, fixed fold order (two-stage folds included): the same call twice gives the same bits."""
    from varsens_amd import _native
    if k == 3:
        d = make_design(ctx, 3, n, 1, np.full(3, -np.pi), np.full(3, np.pi))
        obj = ctx.objective_create(_native.OBJ_ISHIGAMI, 3, 1, np.array([7.0, 0.1]))
    else:
        d = make_design(ctx, k, n)
        obj = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, np.linspace(0.0, 9.0, k))
    nacc, ng = _native.acc_len(k, 1, 1), _native.gram_len(k)
    out = []
    for _ in range(3):
        acc, g = ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
        ctx.eval_reduce_first_second_order(d, obj, 0, n, acc, g)
        out.append((ctx.download(acc, (nacc,)), ctx.download(g, (2, k, k))))
    for a, g in out[1:]:
        assert np.array_equal(a, out[0][0]) and np.array_equal(g, out[0][1])


def test_fused_both_orders_coverage(ctx):
    from varsens_amd import _native
    d = make_design(ctx, 129, 100)
    assert not ctx.eval_reduce_both_supported(d, ctx.objective_create(_native.OBJ_LINEAR, 129, 1, np.ones(129)))   # k > 128
    for kk in (33, 48, 49, 65, 112, 113, 128):
        assert ctx.eval_reduce_both_supported(make_design(ctx, kk, 100), ctx.objective_create(_native.OBJ_LINEAR, kk, 1, np.ones(kk)))
    d8 = make_design(ctx, 8, 100)
    # several outputs (round 4): 2 .. 8 up to 32 index rows, k >= 16 up to 64, 2 or 4 up to 128 (tests/test_gpu_scan.py holds their parity)
    assert ctx.eval_reduce_both_supported(d8, ctx.objective_create(_native.OBJ_LINEAR, 8, 2, np.ones(16)))
    assert ctx.eval_reduce_both_supported(d8, ctx.objective_create(_native.OBJ_LINEAR, 8, 4, np.ones(32)))
    assert not ctx.eval_reduce_both_supported(d8, ctx.objective_create(_native.OBJ_LINEAR, 8, 8, np.ones(64)))    # 64 rows, k < 16
    assert not ctx.eval_reduce_both_supported(d8, ctx.objective_create(_native.OBJ_LINEAR, 8, 9, np.ones(72)))    # nine outputs
    assert not ctx.eval_reduce_both_supported(d8, ctx.objective_create(_native.OBJ_LINEAR, 8, 16, np.ones(128)))  # 16 outputs
    d30 = make_design(ctx, 30, 100)
    assert ctx.eval_reduce_both_supported(d30, ctx.objective_create(_native.OBJ_LINEAR, 30, 4, np.ones(120)))
    assert ctx.eval_reduce_both_supported(d30, ctx.objective_create(_native.OBJ_LINEAR, 30, 3, np.ones(90)))       # 3 outputs, 90 rows
    d40 = make_design(ctx, 40, 100)
    assert not ctx.eval_reduce_both_supported(d40, ctx.objective_create(_native.OBJ_LINEAR, 40, 3, np.ones(120)))  # 3 outputs > 112 rows
    assert not ctx.eval_reduce_both_supported(d30, ctx.objective_create(_native.OBJ_LINEAR, 30, 5, np.ones(150)))  # > 128 rows
    assert not ctx.eval_reduce_both_supported(make_design(ctx, 3, 100), ctx.objective_create(_native.OBJ_INVERT, 3, 3, None))
    lin = ctx.objective_create(_native.OBJ_LINEAR, 8, 1, np.ones(8))
    assert ctx.eval_reduce_both_supported(d8, lin)
    acc, g = ctx.alloc(8 * _native.acc_len(8, 1, 1)), ctx.alloc(8 * _native.gram_len(8))
    with pytest.raises(_native.VarsensHipError):
        ctx.eval_reduce_first_second_order(make_design(ctx, 129, 100), ctx.objective_create(_native.OBJ_LINEAR, 129, 1, np.ones(129)), 0, 100, acc, g)
    # a Sobol design and raw points go through the general column code
    from varsens_amd.saltelli import sobol_directions
    n = 3000
    ds = ctx.design_create(_native.GEN_SOBOL, 8, n, 4096, 0, None, None, orc.seed1_permutation(n).astype(np.uint32), None, sobol_directions(8))
    f = ctx.alloc(8 * 18 * n)
    ctx.eval(ds, lin, 0, n, f)
    acc1, g1 = ctx.alloc(8 * _native.acc_len(8, 1, 1)), ctx.alloc(8 * _native.gram_len(8))
    ctx.reduce_first_second_order(f, n, 8, 1, acc1, g1)
    ctx.eval_reduce_first_second_order(ds, lin, 0, n, acc, g)
    np.testing.assert_allclose(ctx.download(g, (2, 8, 8)), ctx.download(g1, (2, 8, 8)), rtol=1e-12)
    n8 = _native.acc_len(8, 1, 1)
    np.testing.assert_allclose(ctx.download(acc, (n8,))[4:4 + 35], ctx.download(acc1, (n8,))[4:4 + 35], rtol=1e-12)


def sobol_design(ctx, k, n, first, mode=0, lo=None, hi=None):
    from varsens_amd import _native
    from varsens_amd.saltelli import sobol_directions
    dirs = sobol_directions(k)
    perm = orc.seed1_permutation(n).astype(np.uint32)
    return ctx.design_create(_native.GEN_SOBOL, k, n, first, mode, lo, hi, perm, None, dirs), dirs, perm


# first sequence index: the QuantLib tool's 4096 skip (quantlib/sobolGen.cpp:50); one where the indices of a 1000-row
# design cross 2^24 (the seventh and eighth nibble of the Gray code come into use) at a first index that is no multiple
# of 64; one beyond 2^28
SOBOL_FIRSTS = [4096, (1 << 24) - 777, (1 << 28) + 5]


@pytest.fixture(scope="module")
def general_ctx():
    """A context whose scan kernels run the general column code for every design (VS_SCAN_GENERAL=1): Sobol designs
    with identity / linear scaling otherwise take the instance compiled for them alone."""
    import os
    from varsens_amd import _native
    old = {v: os.environ.get(v) for v in ("VS_TUNING", "VS_SCAN_GENERAL")}
    os.environ.update(VS_TUNING="1", VS_SCAN_GENERAL="1")
    try:
        c = _native.Context(0)
    finally:
        for v, x in old.items():
            if x is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = x
    yield c
    c.close()


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("first", SOBOL_FIRSTS)
@pytest.mark.parametrize("k", [4, 16, 24, 32, 48, 64, 96, 128, 200])
def test_sobol_points_through_the_evaluation_kernels(ctx, general_ctx, k, first, general):
    """Round 5: the scan kernels take Sobol points from nibble tables in LDS (vs_sobol_points_lds).  A linear objective
    with one unit weight per output hands the coordinates themselves back as f-values -- every block of the design (M_1,
    M_2 behind the seed-1 shuffle, N_j, N_nj: varsens/saltelli.py:100-125) bit for bit the oracle's Gray-code XOR
    (quantlib/sobolGen.cpp:47-60 is the reference's generator of this sequence), for every kernel shape (columns per wave
    x waves: k = 4 .. 200), a ragged last block and the three index ranges above."""
    from varsens_amd import _native
    ctx = general_ctx if general else ctx     # the column code for Sobol alone / the general one
    n = 1000
    d, dirs, perm = sobol_design(ctx, k, n, first)
    cols = sorted(set(int(c) for c in np.linspace(0, k - 1, min(k, 6))))
    l = len(cols)
    w = np.zeros((l, k))
    w[np.arange(l), cols] = 1.0
    obj = ctx.objective_create(_native.OBJ_LINEAR, k, l, w)
    f = ctx.alloc(8 * 2 * (k + 1) * n * l)
    ctx.eval(d, obj, 0, n, f)
    F = ctx.download_blocks(f, k, l, n)                       # [2k+2][n][l]
    m1 = orc.sobol(k, first, n, dirs)
    seq2 = orc.sobol(k, first + n, n, dirs)
    m2 = seq2[perm.astype(np.int64)]
    same(F[0], m1[:, cols])
    same(F[1], m2[:, cols])
    for j in range(k):
        nj, nnj = m2[:, cols].copy(), m1[:, cols].copy()
        if j in cols:
            o = cols.index(j)
            nj[:, o], nnj[:, o] = m1[:, j], m2[:, j]
        same(F[2 + j], nj)                                    # M_2 with column j of M_1 (saltelli.py:112-125)
        same(F[2 + k + j], nnj)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("first", SOBOL_FIRSTS)
@pytest.mark.parametrize("kind,k,l", [("lin", 4, 1), ("g", 16, 1), ("poly", 24, 1), ("g", 32, 1), ("lin", 48, 1), ("g", 64, 1),
                                      ("poly", 96, 1), ("g", 128, 1), ("lin", 16, 2), ("g", 20, 3), ("lin", 32, 4)])
def test_sobol_designs_through_the_fused_kernels(ctx, kind, k, l, first, mode):
    """The fused kernels (evaluate + reduce, no f-values) on Sobol designs against evaluation + reduction of the same
    design: record and Gram sums to summation order.  The first-order kernel of every shape (all tables in LDS up to 96
    columns, two columns at a time beyond), the both-orders kernels of 16 / 32 / 64 / 128 index rows with one output and
    with several (32 x 4: the instance that has no LDS left and keeps the byte tables in memory).  mode 1: linear scaling
    (one output: the column code compiled for Sobol alone), 2: power scaling (the general column code)."""
    from varsens_amd import _native
    n = 3000
    rng = np.random.RandomState(k + l)
    code = {"g": _native.OBJ_G_FUNCTION, "lin": _native.OBJ_LINEAR, "poly": _native.OBJ_POLYNOMIAL}[kind]
    params = {"g": rng.rand(l, k) * 5, "lin": rng.rand(l, k) + 0.5, "poly": 1.0 / (1.0 + rng.rand(l, k))}[kind]
    lo, hi = np.linspace(0.0, 0.2, k), np.linspace(0.8, 1.0, k)   # (a product of 128 terms stays of order 1)
    if mode == 2:
        lo = lo + 0.05
    d, _, _ = sobol_design(ctx, k, n, first, mode, lo, hi)
    obj = ctx.objective_create(code, k, l, params)
    nacc, ng = _native.acc_len(k, l, 1), _native.gram_len(k * l)
    f = ctx.alloc(8 * 2 * (k + 1) * n * l)
    ctx.eval(d, obj, 0, n, f)
    acc1, acc2, acc3 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    g1, g3 = ctx.alloc(8 * ng), ctx.alloc(8 * ng)
    ctx.reduce_first_second_order(f, n, k, l, acc1, g1)
    r1 = ctx.download(acc1, (nacc,))
    per = _native.acc_per_output(k)
    sums = slice(4, 4 + per * l)
    if ctx.eval_reduce_supported(d, obj):
        ctx.eval_reduce_first_order(d, obj, 0, n, acc2)
        r2 = ctx.download(acc2, (nacc,))
        assert r2[0] == n and r2[1] == 0
        np.testing.assert_allclose(r2[sums], r1[sums], rtol=1e-11)
    else:
        assert l > 1 and k * l > 128
    if not ctx.eval_reduce_both_supported(d, obj):
        assert mode == 2 and l > 1   # (power scaling with several outputs at 33 .. 64 index rows: two kernels)
        return
    ctx.eval_reduce_first_second_order(d, obj, 0, n, acc3, g3)
    r3 = ctx.download(acc3, (nacc,))
    np.testing.assert_allclose(r3[sums], r1[sums], rtol=1e-11)
    G1, G3 = ctx.download(g1, (2, k * l, k * l)), ctx.download(g3, (2, k * l, k * l))
    np.testing.assert_allclose(G3, G1, rtol=1e-11, atol=1e-12 * np.abs(G1).max())


def test_variance_with_outlier_in_first_value(ctx):
    """The variance shift is a leading-sample mean, not the first value: an extreme first value must not
    cost accuracy (numpy's two-pass variance is the reference, saltelli.py:583)."""
    rng = np.random.RandomState(13)
    blocks = 5.0 + rng.randn(2 * 3 + 2, 50000, 1)
    blocks[0, 0, 0] = 1e9
    rec, fin = reduce_dev(ctx, blocks, 50000)
    want = np.var(np.concatenate([blocks[0], blocks[1]]), axis=0, ddof=1)
    np.testing.assert_allclose(fin[1], want, rtol=1e-9)
    blocks[0, 0, 0] = 5.0
    blocks[0, 70, 0] = -1e7          # outlier inside the leading sample
    rec, fin = reduce_dev(ctx, blocks, 50000)
    want = np.var(np.concatenate([blocks[0], blocks[1]]), axis=0, ddof=1)
    np.testing.assert_allclose(fin[1], want, rtol=1e-10)


def test_reduce_very_wide_k(ctx):
    """k = 600 > 512: the LDS accumulator is split over blockIdx.y; k = 200 second order = 7 super-blocks."""
    rng = np.random.RandomState(600)
    k, rows = 600, 301
    blocks = 1.0 + rng.rand(2 * k + 2, rows, 1)
    rec, fin = reduce_dev(ctx, blocks, rows)
    check_against_exact(rec, fin, blocks, rows)
    from varsens_amd import _native
    k2, rows2 = 200, 150
    b2 = rng.randn(2 * k2 + 2, rows2, 1)
    f = ctx.upload_blocks(b2)
    gram = ctx.alloc(8 * 2 * k2 * k2)
    ctx.reduce_second_order(f, rows2, k2, 1, gram)
    G = ctx.download(gram, (2, k2, k2))
    C, D = b2[2:2 + k2, :, 0], b2[2 + k2:, :, 0]
    np.testing.assert_allclose(G[0], (C + D) @ (C + D).T, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(G[1], (C - D) @ (C - D).T, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("k,l", [(5, 1), (7, 3), (40, 1)])
def test_finish_second_order_formulas(ctx, k, l):
    """vs_finish_second_order alone: from uploaded S S^T, T T^T and a first-order block to the reference's
    sens_2 / sens_2n (varsens/saltelli.py:612-622, E_2 and var_y broadcast over the LAST output axis)."""
    from varsens_amd import _native
    rng = np.random.RandomState(k * 10 + l)
    K, n = k * l, 777
    C = 2.0 + rng.rand(K, n)
    D = 2.0 + rng.rand(K, n)
    S, T = C + D, C - D
    gram = ctx.upload(np.concatenate([(S @ S.T).ravel(), (T @ T.T).ravel()]))
    first = np.zeros((2 + 4 * k, l))
    first[0] = 4.0 + rng.rand(l)          # E_2
    first[1] = 0.5 + rng.rand(l)          # var_y
    s2, s2n = ctx.alloc(8 * K * K), ctx.alloc(8 * K * K)
    ctx.finish_second_order(gram, ctx.upload(first), n, k, l, s2, s2n)
    div = 2.0 * (n - 1)
    want2 = ((D @ C.T + C @ D.T) / div).reshape(k, l, k, l)
    want2 = (want2 - first[0]) / first[1]
    want2n = ((D @ D.T + C @ C.T) / div).reshape(k, l, k, l)
    want2n = (want2n - first[0]) / first[1]
    np.testing.assert_allclose(ctx.download(s2, (k, l, k, l)), want2, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ctx.download(s2n, (k, l, k, l)), want2n, rtol=1e-12, atol=1e-12)


def test_abi_argument_errors(ctx):
    """Bad arguments come back as VS_ERR_ARG with a message, never as a crash."""
    from varsens_amd import _native
    E = _native.VarsensHipError
    with pytest.raises(E, match="out of range"):
        ctx.design_create(_native.GEN_HALTON, 0, 10, 1)
    with pytest.raises(E, match="unknown generator"):
        ctx.design_create(7, 3, 10, 1)
    with pytest.raises(E, match="bounds are required"):
        ctx.design_create(_native.GEN_HALTON, 3, 10, 1, _native.SCALE_LINEAR)
    with pytest.raises(E, match="not a row of the second half"):
        ctx.design_create(_native.GEN_HALTON, 3, 4, 1, 0, None, None, np.array([0, 1, 2, 9], dtype=np.uint32))
    d = make_design(ctx, 3, 16)
    with pytest.raises(E, match="parameters"):
        ctx.objective_create(_native.OBJ_LINEAR, 3, 1, [1.0, 2.0])
    with pytest.raises(E, match="unknown objective"):
        ctx.objective_create(99, 3, 1, [1.0, 2.0, 3.0])
    o4 = ctx.objective_create(_native.OBJ_LINEAR, 4, 1, np.ones(4))
    f = ctx.alloc(8 * 8 * 16)
    with pytest.raises(E, match="objective has k=4, design has k=3"):
        ctx.eval(d, o4, 0, 16, f)
    o3 = ctx.objective_create(_native.OBJ_LINEAR, 3, 1, np.ones(3))
    with pytest.raises(E, match="exceed n"):
        ctx.eval(d, o3, 10, 7, f)
    acc = ctx.alloc(8 * _native.acc_len(3, 1, 2))
    with pytest.raises(E, match="rank 2 not in"):
        ctx.reduce_first_order(f, 16, 3, 1, acc, 0, 2, 2)
    with pytest.raises(E, match="exceed 2n"):
        ctx.sample_flat(d, 120, 20, f)
    with pytest.raises(E, match="not been called"):
        ctx.comm_allreduce(acc, 4)


def test_very_large_k(ctx):
    """k = 1000: Halton bases up to 7919, scratch-tile evaluation, first order split over blockIdx.y."""
    from varsens_amd import _native
    k, n = 1000, 70
    c = 1.0 / np.arange(1, k + 1)
    d = make_design(ctx, k, n)
    f, got = eval_dev(ctx, d, _native.OBJ_LINEAR, k, 1, c, n)
    S = orc.build_sample(k, n)
    want = oracle_blocks(orc.evaluate_rows(lambda X: orc.linear_rows(X, c), *S))
    same(got, want)
    rec, fin = reduce_dev(ctx, got, n)
    check_against_exact(rec, fin, got, n)


def test_design_with_a_slice_of_the_shuffle(ctx):
    """A rank of a multi-GPU job uploads only the seed-1 shuffle entries of its own rows (vs_design_desc.perm_begin /
    perm_count): same values as the whole design on those rows, other rows refused."""
    from varsens_amd import _native
    k, n = 9, 5001
    begin, count = 1250, 1250
    perm = orc.seed1_permutation(n).astype(np.uint32)
    whole = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, perm)
    part = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, perm[begin:begin + count],
                             perm_begin=begin)
    a = np.linspace(0.0, 4.0, k)
    _, want = eval_dev(ctx, whole, _native.OBJ_G_FUNCTION, k, 1, a, n, begin=begin, count=count)
    _, got = eval_dev(ctx, part, _native.OBJ_G_FUNCTION, k, 1, a, n, begin=begin, count=count)
    same(got, want)
    _, got = eval_dev(ctx, part, _native.OBJ_INVERT, k, k, None, n, begin=begin + 7, count=100)       # chain kernel
    _, want = eval_dev(ctx, whole, _native.OBJ_INVERT, k, k, None, n, begin=begin + 7, count=100)
    same(got, want)
    obj = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, a)
    nacc = _native.acc_len(k, 1, 1)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    ctx.eval_reduce_first_order(whole, obj, begin, count, acc1)
    ctx.eval_reduce_first_order(part, obj, begin, count, acc2)
    same(ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,)))
    f = ctx.alloc(8 * 2 * (k + 1) * 10)
    with pytest.raises(_native.VarsensHipError, match="shuffle slice"):
        ctx.eval(part, obj, begin - 1, 10, f)
    with pytest.raises(_native.VarsensHipError, match="shuffle slice"):
        ctx.eval_reduce_first_order(part, obj, begin + count - 5, 10, acc1)
    with pytest.raises(_native.VarsensHipError, match="whole shuffle"):
        ctx.sample_flat(part, 0, 10, f)
    with pytest.raises(_native.VarsensHipError, match="exceeds n|perm must hold"):
        ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, perm[:100], perm_begin=n - 50)


def test_device_halton_against_scipy(ctx):
    """vs_halton_fill against scipy's unscrambled Halton directly (not through the oracle): <= 2 ulp, k = 128,
    indices 1.. and across a large offset.  ghalton itself is unavailable: see the oracle header."""
    from scipy.stats import qmc
    for k, start, count in ((128, 1, 4096), (32, 123457, 3000)):
        eng = qmc.Halton(d=k, scramble=False)
        eng.fast_forward(start)
        ref = eng.random(count)
        got = ctx.download(ctx.halton_fill(k, start, count), (count, k))
        d = np.abs(got - ref) / np.spacing(np.maximum(got, ref))
        assert d.max() <= (2.0 if start < 100000 else 4.0) and np.percentile(d, 99) <= 2.0 and (d == 0).mean() > 0.45


# The fused kernels form Halton numerators in three instructions per digit (vs_radical_inverse_pair_horner) where
# (largest index of the design) * b^(digits - 1) < 2^53 holds for the column -- the host's VS_BASE_HORNER -- and fall
# back to the digit loop column by column where it does not (general column code), so designs on either side of the
# condition, and designs with columns on both sides, must give the oracle's bits.  k = 5: base 3 is the first column to
# lose the flag, when the last index reaches 3^17 (then 17 -> 18 digits, top weight 3^17: 3^17 * 3^17 > 2^53).
_K53, _N53 = 5, 96
_LAST_OK = 3 ** 17 - 1
_DISCARDS_53 = [("every column in the three-instruction form (last index 3^17 - 1)", _LAST_OK - 2 * _N53 + 1 - (1 + 20 * _K53)),
                ("base 3 back on the digit loop (last index 3^17)", _LAST_OK - 2 * _N53 + 2 - (1 + 20 * _K53)),
                ("bases 2 and 3 on the digit loop (last index 2^27 + 5)", 2 ** 27 + 5 - 2 * _N53 + 1 - (1 + 20 * _K53)),
                ("far beyond (last index 3e9)", 3_000_000_000 - 2 * _N53 + 1 - (1 + 20 * _K53))]


@pytest.mark.parametrize("label,discard", _DISCARDS_53, ids=[c[0] for c in _DISCARDS_53])
@pytest.mark.parametrize("scaled", [False, True], ids=["identity", "linear"])
def test_designs_around_the_exactness_bound_of_the_short_digit_step(ctx, label, discard, scaled):
    from varsens_amd import _native
    k, n = _K53, _N53
    lo, hi = np.linspace(-2.0, 1.0, k), np.linspace(3.0, 9.0, k)
    if scaled:
        d = make_design(ctx, k, n, _native.SCALE_LINEAR, lo, hi, discard=discard)
        S = orc.build_sample(k, n, lambda x: orc.scale_linear(x, lo, hi), discard=discard)
    else:
        d = make_design(ctx, k, n, discard=discard)
        S = orc.build_sample(k, n, discard=discard)
    for kind, name, params in ((_native.OBJ_G_FUNCTION, "g_function", np.array([0.0, 0.5, 3.0, 9.0, 99.0])),
                               (_native.OBJ_LINEAR, "linear", np.arange(1.0, k + 1))):
        f, got = eval_dev(ctx, d, kind, k, 1, params, n)
        F = orc.evaluate_engine_order(name, params, S[0], S[1])
        same(got, oracle_blocks(F))                                   # vs_eval: the digit loop in every design
        obj = ctx.objective_create(kind, k, 1, params)
        assert ctx.eval_reduce_supported(d, obj) and ctx.eval_reduce_both_supported(d, obj)
        nacc, ng = _native.acc_len(k, 1, 1), _native.gram_len(k)
        acc, acc2, gram = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
        ctx.eval_reduce_first_order(d, obj, 0, n, acc)                # fused kernels: the short form where the column allows
        ctx.eval_reduce_first_second_order(d, obj, 0, n, acc2, gram)
        ex = orc.exact_accumulators(*orc.evaluate_scan(name, params, S[0], S[1], fused=True))
        for r in (ctx.download(acc, (nacc,)), ctx.download(acc2, (nacc,))):
            np.testing.assert_allclose(r[4 + 3:4 + 3 + k], ex["ac"][:, 0], rtol=1e-13)
            np.testing.assert_allclose(r[4 + 3 + k:4 + 3 + 2 * k], ex["bd"][:, 0], rtol=1e-13)
        C, D = np.asarray(F[2])[:, :, 0], np.asarray(F[3])[:, :, 0]
        G = ctx.download(gram, (2, k, k))
        np.testing.assert_allclose(G[0], (C + D) @ (C + D).T, rtol=1e-12)
        np.testing.assert_allclose(G[1], (C - D) @ (C - D).T, rtol=1e-12, atol=1e-12 * np.abs(G[0]).max())


def test_library_supplied_seed1_shuffle_and_its_cache(ctx):
    """VS_DESIGN_SEED1_SHUFFLE (ABI 7): the design made from the library's own shuffle evaluates to the bits of the one
    made from the uploaded permutation -- whole and as a rank's slice, an empty slice included --; the device copy is
    shared by later designs of the same (n, slice) and outlives the designs that used it; more (n, slice) pairs than the
    cache holds, alive at once and released in any order, stay correct."""
    from varsens_amd import _native
    k = 5
    params = np.array([0.0, 0.5, 3.0, 9.0, 99.0])

    def blocks(design, n, begin, count):
        _, got = eval_dev(ctx, design, _native.OBJ_G_FUNCTION, k, 1, params, n, begin=begin, count=count)
        return got

    alive = []
    for n in (64, 1000, 1001, 4097, 70000, 333):        # six sizes: more than the four copies kept unreferenced
        perm = orc.seed1_permutation(n).astype(np.uint32)
        want = blocks(ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, perm), n, 0, n)
        d1 = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, None, seed1_shuffle=(0, n))
        same(blocks(d1, n, 0, n), want)
        d2 = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, None, seed1_shuffle=(0, n))
        d1.free()                                        # the shared copy stays for d2
        same(blocks(d2, n, 0, n), want)
        b, c = n // 3, n // 2                            # a rank's slice: only its rows evaluate
        ds = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, None, seed1_shuffle=(b, c))
        same(blocks(ds, n, b, c), want[:, b:b + c])
        with pytest.raises(_native.VarsensHipError, match="shuffle slice"):
            blocks(ds, n, 0, n)
        de = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, None, seed1_shuffle=(n, 0))
        de.free()
        alive += [(d2, n, want), (ds, n, want[:, b:b + c], b, c)]
    for item in alive[::2] + alive[1::2]:                # every design still evaluates, then goes, in mixed order
        if len(item) == 3:
            same(blocks(item[0], item[1], 0, item[1]), item[2])
        else:
            same(blocks(item[0], item[1], item[3], item[4]), item[2])
        item[0].free()
    with pytest.raises(_native.VarsensHipError, match="exclude each other"):
        ctx.design_create(_native.GEN_HALTON, k, 64, orc.halton_skip(k), 0, None, None,
                          orc.seed1_permutation(64).astype(np.uint32), seed1_shuffle=(0, 64))
