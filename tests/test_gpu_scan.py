"""The segmented-scan kernels (varsens_amd/csrc/vs_device_scan.h, vs_scan.hip) against the oracle, through
the C ABI.  GPU only.

* evaluation (vs_eval, k >= 33 by itself; forced at small k through VS_EVAL_MODE=2): bit-exact against the
  oracle's restatement of the kernels' association order (orc.evaluate_scan), rounding-close (<= 6e-15
  relative) to the strict left-to-right folds the chain kernel and the reference's numpy.prod produce;
* fused evaluate + first-order reduce (vs_eval_reduce_first_order): its accumulator record against the
  exactly rounded sums of the oracle's values in the fused kernels' order, and against the two-kernel path
  (<= 2e-13 relative), for k in {33, 64, 128}, l in {1, 2, 3}, Ishigami and separable source objectives.
"""
import numpy as np
import pytest

from oracle import saltelli_oracle as orc

pytestmark = pytest.mark.gpu

KINDS = {"g_function": 1, "linear": 3, "polynomial": 4}
STRICT = {"g_function": orc.g_function_rows, "linear": orc.linear_rows, "polynomial": orc.polynomial_rows}


@pytest.fixture(scope="module")
def ctx():
    from varsens_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def scan_ctx():
    """A context whose vs_eval takes the scan kernels at every k they cover (VS_EVAL_MODE=2, read once at
    context creation)."""
    import os
    from varsens_amd import _native
    os.environ["VS_EVAL_MODE"] = "2"
    os.environ["VS_TUNING"] = "1"
    try:
        c = _native.Context(0)
    finally:
        del os.environ["VS_EVAL_MODE"]
        del os.environ["VS_TUNING"]
    yield c
    c.close()


def same(a, b):
    assert a.shape == b.shape
    assert np.array_equal(a, b, equal_nan=True), "max |d| = %g" % np.nanmax(np.abs(a - b))


def params_for(kind, k, l, seed):
    rng = np.random.RandomState(seed)
    if kind == "g_function":
        P = rng.rand(l, k) * 9.0
    elif kind == "linear":
        P = np.arange(1, k + 1, dtype=float)[None, :] * (1.0 + np.arange(l))[:, None]
    else:
        P = (1.0 / (1.0 + np.arange(k) / 8.0))[None, :] * (1.0 - 0.25 * np.arange(l))[:, None]
    return np.ascontiguousarray(P)


def design(ctx, k, n, mode=0, lo=None, hi=None, raw=None, generator=None, dirs=None, first=None):
    from varsens_amd import _native
    perm = orc.seed1_permutation(n).astype(np.uint32)
    if raw is not None:
        return ctx.design_create(_native.GEN_RAW, k, n, 0, mode, lo, hi, perm, raw)
    if generator == "sobol":
        return ctx.design_create(_native.GEN_SOBOL, k, n, 4096 if first is None else first, mode, lo, hi, perm, None, dirs)
    return ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k) if first is None else first, mode, lo, hi, perm)


def device_blocks(ctx, d, kind, k, l, P, n, begin=0, count=None):
    count = n if count is None else count
    obj = ctx.objective_create(KINDS[kind], k, l, P)
    f = ctx.alloc(8 * 2 * (k + 1) * count * l)
    ctx.eval(d, obj, begin, count, f)
    return f, obj, ctx.download_blocks(f, k, l, count)


def blocks_of(F):
    return np.concatenate([F[0][None], F[1][None], F[2], F[3]], axis=0)


def strict_blocks(kind, P, S):
    return blocks_of(orc.evaluate_rows(lambda X: np.stack([STRICT[kind](X, p) for p in P], axis=-1), *S))


# ------------------------------------------------------------------ evaluation

@pytest.mark.parametrize("kind", ["g_function", "linear", "polynomial"])
@pytest.mark.parametrize("k,n,l", [(33, 300, 1), (40, 1000, 1), (64, 129, 1), (65, 64, 1), (100, 333, 2), (128, 500, 1),
                                   (129, 70, 1), (200, 100, 1), (256, 65, 3)])
def test_scan_eval_default_route(ctx, kind, k, n, l):
    P = params_for(kind, k, l, k + n)
    S = orc.build_sample(k, n)
    _, _, got = device_blocks(ctx, design(ctx, k, n), kind, k, l, P, n)
    same(got, blocks_of(orc.evaluate_scan(kind, P, S[0], S[1])))
    strict = strict_blocks(kind, P, S)
    assert np.max(np.abs(got - strict) / np.maximum(np.abs(strict), 1e-300)) <= 6e-15


def test_g_function_quotient_is_the_correctly_rounded_one(ctx):
    """The scan kernels take (|4x-2| + a) / (1 + a) from a host-rounded reciprocal by Markstein's fma sequence: the
    bits must be those of the division (numpy's, the reference's) for every parameter the host admits -- tiny, huge,
    many-ones significands -- and for a parameter it does not admit (1 + a with an all-ones significand, a = -1, inf)
    that column alone divides (round 4: the objective stays on the scan kernels and the fused paths; until round 3 one
    such parameter sent it to the O(k^2) chain kernels)."""
    from varsens_amd import _native
    k, n = 24, 20000
    a = np.array([0.0, 0.5, 3.0, 9.0, 99.0, 99.0, 1e-12, 1e-300, 1e6, 2.0 ** 52, 1.0 / 3.0, 2.0 / 3.0, 0.1, 0.7,
                  1.0 - 2.0 ** -51, 2.0 ** -53, 7.0 - 2.0 ** -49, 1e15 / 3.0, np.pi, np.e, 0.999999999999, 123456.789,
                  2.0 ** 40 - 0.25, 5e-324])
    assert len(a) == k
    S = orc.build_sample(k, n)
    d = design(ctx, k, n)
    _, obj, got = device_blocks(ctx, d, "g_function", k, 1, a[None, :], n)
    assert ctx.eval_reduce_supported(d, obj)
    same(got, blocks_of(orc.evaluate_scan("g_function", a[None, :], S[0], S[1])))
    for bad in (1.0 - 2.0 ** -52, 7.0 - 2.0 ** -50, -1.0, np.inf):        # 1 + a = 1.11...1b (twice), 0, inf
        b = a.copy()
        b[5] = bad
        _, obj, got = device_blocks(ctx, d, "g_function", k, 1, b[None, :], n)
        assert ctx.eval_reduce_supported(d, obj) and ctx.eval_reduce_both_supported(d, obj)
        with np.errstate(all="ignore"):
            same(got, blocks_of(orc.evaluate_scan("g_function", b[None, :], S[0], S[1])))
            if np.isfinite(bad) and bad != -1.0:     # the fused kernels on the same objective: finite sums
                nacc = _native.acc_len(k, 1, 1)
                acc = ctx.alloc(8 * nacc)
                ctx.eval_reduce_first_order(d, obj, 0, n, acc)
                check_record(ctx.download(acc, (nacc,)), orc.evaluate_scan("g_function", b[None, :], S[0], S[1], fused=True), k, 1)


@pytest.mark.parametrize("kind,k,n,l", [("g_function", 6, 1024, 1), ("g_function", 6, 1000, 2), ("linear", 1, 7, 1),
                                        ("polynomial", 2, 64, 1), ("g_function", 9, 65, 1), ("linear", 16, 1, 1),
                                        ("polynomial", 17, 4097, 1), ("linear", 32, 500, 1), ("g_function", 31, 200, 3)])
def test_scan_eval_forced_at_small_k(scan_ctx, kind, k, n, l):
    P = params_for(kind, k, l, 7 * k + n)
    S = orc.build_sample(k, n)
    _, _, got = device_blocks(scan_ctx, design(scan_ctx, k, n), kind, k, l, P, n)
    same(got, blocks_of(orc.evaluate_scan(kind, P, S[0], S[1])))
    strict = strict_blocks(kind, P, S)
    assert np.max(np.abs(got - strict) / np.maximum(np.abs(strict), 1e-300)) <= 2e-15


def test_scan_eval_row_shards_and_scalings(ctx):
    """A row shard evaluates to the same values; linear and power scaling, raw points and Sobol points go
    through the same kernels (compact generator paths)."""
    from varsens_amd import _native
    from varsens_amd.saltelli import sobol_directions
    k, n = 48, 777
    P = params_for("g_function", k, 1, 3)
    lo, hi = np.linspace(-1.0, 0.5, k), np.linspace(1.0, 3.0, k)
    S = orc.build_sample(k, n, lambda x: orc.scale_linear(x, lo, hi))
    d = design(ctx, k, n, _native.SCALE_LINEAR, lo, hi)
    _, _, whole = device_blocks(ctx, d, "g_function", k, 1, P, n)
    same(whole, blocks_of(orc.evaluate_scan("g_function", P, S[0], S[1])))
    _, _, part = device_blocks(ctx, d, "g_function", k, 1, P, n, begin=130, count=211)
    same(part, whole[:, 130:341])
    # power scaling: lower * exp(x ln r) on the device within two ulp of numpy's pow per coordinate
    plo, phi = np.full(k, 0.5), np.linspace(2.0, 50.0, k)
    Sp = orc.build_sample(k, n, lambda x: orc.scale_power(x, plo, phi))
    _, _, got = device_blocks(ctx, design(ctx, k, n, _native.SCALE_POWER, plo, phi), "polynomial", k, 1,
                              params_for("polynomial", k, 1, 0), n)
    want = blocks_of(orc.evaluate_scan("polynomial", params_for("polynomial", k, 1, 0), Sp[0], Sp[1]))
    np.testing.assert_allclose(got, want, rtol=1e-13)   # (a product of k terms 1 + beta (x - 1/2): the ulps add up)
    # raw points
    rng = np.random.RandomState(5)
    raw = rng.rand(2 * n, k)
    Sr = orc.build_sample(k, n, raw=raw)
    _, _, got = device_blocks(ctx, design(ctx, k, n, raw=raw), "linear", k, 1, params_for("linear", k, 1, 0), n)
    same(got, blocks_of(orc.evaluate_scan("linear", params_for("linear", k, 1, 0), Sr[0], Sr[1])))
    # Sobol points
    dirs = sobol_directions(k)
    pts = orc.sobol(k, 4096, 2 * n, dirs)
    Ss = orc.build_sample(k, n, raw=pts)
    _, _, got = device_blocks(ctx, design(ctx, k, n, generator="sobol", dirs=dirs), "g_function", k, 1, P, n)
    same(got, blocks_of(orc.evaluate_scan("g_function", P, Ss[0], Ss[1])))


# ------------------------------------------------------------------ fused evaluate + reduce

def record_parts(rec, k, l):
    from varsens_amd import _native
    per = _native.acc_per_output(k)
    body = rec[4:4 + l * per].reshape(l, per)
    return dict(rows=rec[0], sab=body[:, 0], sa=body[:, 1], sb=body[:, 2], ac=body[:, 3:3 + k].T,
                bd=body[:, 3 + k:3 + 2 * k].T, ad=body[:, 3 + 2 * k:3 + 3 * k].T, bc=body[:, 3 + 3 * k:3 + 4 * k].T,
                slots=rec[4 + l * per:])


def check_record(rec, F, k, l, rtol=1e-13):
    ex = orc.exact_accumulators(*F)
    got = record_parts(rec, k, l)
    assert got["rows"] == F[0].shape[0]
    for name in ("sab", "sa", "sb", "ac", "bd", "ad", "bc"):
        np.testing.assert_allclose(got[name], ex[name], rtol=rtol, err_msg=name)
    cnt, mean, m2 = got["slots"][0], got["slots"][1:1 + l], got["slots"][1 + l:1 + 2 * l]
    assert cnt == 2 * F[0].shape[0]
    np.testing.assert_allclose(mean, ex["mean"], rtol=1e-13)
    np.testing.assert_allclose(m2, ex["m2"], rtol=1e-11)       # sum of squared deviations: conditioned by sigma/mean


@pytest.mark.parametrize("kind", ["g_function", "linear", "polynomial"])
@pytest.mark.parametrize("k,n,l", [(6, 5000, 1), (1, 64, 1), (16, 333, 1), (31, 1, 1), (32, 20001, 1), (33, 4000, 1),
                                   (64, 3000, 1), (128, 2049, 1), (33, 700, 2), (64, 650, 3), (128, 300, 2), (7, 900, 3)])
def test_fused_record(ctx, kind, k, n, l):
    from varsens_amd import _native
    P = params_for(kind, k, l, 11 * k + n + l)
    d = design(ctx, k, n)
    f, obj, _ = device_blocks(ctx, d, kind, k, l, P, n)
    nacc = _native.acc_len(k, l, 1)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    assert ctx.eval_reduce_supported(d, obj)
    ctx.reduce_first_order(f, n, k, l, acc1)
    ctx.eval_reduce_first_order(d, obj, 0, n, acc2)
    r1, r2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    # against exact sums of the oracle's values in the fused kernels' association order
    S = orc.build_sample(k, n)
    check_record(r2, orc.evaluate_scan(kind, P, S[0], S[1], fused=True), k, l)
    # against the two-kernel path (whose f-values differ by the association order of the folds only)
    per = _native.acc_per_output(k)
    np.testing.assert_allclose(r2[4:4 + l * per], r1[4:4 + l * per], rtol=2e-13)
    if n > 1:
        out1, out2 = ctx.alloc(8 * _native.first_order_len(k, l)), ctx.alloc(8 * _native.first_order_len(k, l))
        ctx.finish_first_order(acc1, n, k, l, out1)
        ctx.finish_first_order(acc2, n, k, l, out2)
        a, b = ctx.download(out1, (2 + 4 * k, l)), ctx.download(out2, (2 + 4 * k, l))
        np.testing.assert_allclose(b[:2 + 2 * k], a[:2 + 2 * k], rtol=1e-12)
        np.testing.assert_allclose(b[2 + 2 * k:], a[2 + 2 * k:], rtol=1e-9, atol=1e-11)


def test_fused_shard_slot_and_linear_scaling(ctx):
    """A rank's shard: rows [begin, begin+count) of a bigger design, its own variance slot of 3."""
    from varsens_amd import _native
    k, n, l = 40, 5000, 1
    lo, hi = np.full(k, -2.0), np.linspace(1.0, 4.0, k)
    P = params_for("polynomial", k, l, 1)
    d = design(ctx, k, n, _native.SCALE_LINEAR, lo, hi)
    obj = ctx.objective_create(KINDS["polynomial"], k, l, P)
    begin, count = 1234, 2001
    nacc = _native.acc_len(k, l, 3)
    acc = ctx.alloc(8 * nacc)
    ctx.eval_reduce_first_order(d, obj, begin, count, acc, rank=1, max_ranks=3)
    rec = ctx.download(acc, (nacc,))
    S = orc.build_sample(k, n, lambda x: orc.scale_linear(x, lo, hi))
    F = orc.evaluate_scan("polynomial", P, S[0][begin:begin + count], S[1][begin:begin + count], fused=True)
    parts = record_parts(rec, k, l)
    slots = parts["slots"].reshape(3, 3)
    assert slots[0, 0] == 0 and slots[2, 0] == 0 and slots[1, 0] == 2 * count
    parts["slots"] = slots[1]
    ex = orc.exact_accumulators(*F)
    for name in ("sab", "sa", "sb", "ac", "bd", "ad", "bc"):
        np.testing.assert_allclose(parts[name], ex[name], rtol=1e-13, err_msg=name)
    np.testing.assert_allclose(slots[1, 1], ex["mean"][0], rtol=1e-13)


def test_fused_ishigami(ctx):
    from varsens_amd import _native
    n = 100003
    lo, hi = np.full(3, -np.pi), np.full(3, np.pi)
    d = design(ctx, 3, n, _native.SCALE_LINEAR, lo, hi)
    obj = ctx.objective_create(_native.OBJ_ISHIGAMI, 3, 1, [7.0, 0.1])
    assert ctx.eval_reduce_supported(d, obj)
    nacc = _native.acc_len(3, 1, 1)
    f = ctx.alloc(8 * 8 * n)
    ctx.eval(d, obj, 0, n, f)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    ctx.reduce_first_order(f, n, 3, 1, acc1)
    ctx.eval_reduce_first_order(d, obj, 0, n, acc2)
    r1, r2 = ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,))
    np.testing.assert_allclose(r2[:4 + 15], r1[:4 + 15], rtol=2e-13)       # same operations per value, other summation order
    np.testing.assert_allclose(r2[4 + 15:], r1[4 + 15:], rtol=1e-11)
    out = ctx.alloc(8 * _native.first_order_len(3, 1))
    ctx.finish_first_order(acc2, n, 3, 1, out)
    fin = ctx.download(out, (14, 1))
    V, S, ST = orc.ishigami_truth()
    np.testing.assert_allclose(fin[8:11, 0], S, atol=2e-2)
    np.testing.assert_allclose(fin[11:14, 0], ST, atol=2e-2)


@pytest.mark.parametrize("k,n", [(6, 3000), (32, 2000), (50, 1500), (128, 600)])
def test_fused_separable_source_objective(ctx, k, n):
    """A user-written separable objective (hiprtc) takes the same scan kernels as the built-in of the same
    formula: identical f-values and an identical fused record."""
    from varsens_amd import _native
    a = np.linspace(0.0, 9.0, k)
    d = design(ctx, k, n)
    user = ctx.objective_create_separable(k, "return (fabs(4.0 * x - 2.0) + p[c]) / (1.0 + p[c]);", False, a)
    builtin = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, a)
    f1, f2 = ctx.alloc(8 * 2 * (k + 1) * n), ctx.alloc(8 * 2 * (k + 1) * n)
    ctx.eval(d, user, 0, n, f1)
    ctx.eval(d, builtin, 0, n, f2)
    same(ctx.download(f1, (2 * k + 2, n)), ctx.download(f2, (2 * k + 2, n)))
    assert ctx.eval_reduce_supported(d, user)
    nacc = _native.acc_len(k, 1, 1)
    acc1, acc2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc)
    ctx.eval_reduce_first_order(d, user, 0, n, acc1)
    ctx.eval_reduce_first_order(d, builtin, 0, n, acc2)
    same(ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,)))
    # and a sum-separable one against the built-in linear objective
    c = 1.0 + np.arange(k)
    lin_user = ctx.objective_create_separable(k, "return p[c] * x;", True, c)
    lin = ctx.objective_create(_native.OBJ_LINEAR, k, 1, c)
    ctx.eval_reduce_first_order(d, lin_user, 0, n, acc1)
    ctx.eval_reduce_first_order(d, lin, 0, n, acc2)
    same(ctx.download(acc1, (nacc,)), ctx.download(acc2, (nacc,)))


def test_fused_is_bit_reproducible(ctx):
    from varsens_amd import _native
    k, n = 64, 40000
    d = design(ctx, k, n)
    obj = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, np.linspace(0, 5, k))
    nacc = _native.acc_len(k, 1, 1)
    acc = ctx.alloc(8 * nacc)
    ctx.eval_reduce_first_order(d, obj, 0, n, acc)
    first = ctx.download(acc, (nacc,))
    ctx.eval_reduce_first_order(d, obj, 0, n, acc)
    same(ctx.download(acc, (nacc,)), first)


# ------------------------------------------------------------------ fused evaluate + BOTH orders, straight against the oracle

def gram_expected(F):
    """S S^T and T T^T (S = C + D, T = C - D; index row u = output * k + column) of ORACLE f-values, in numpy."""
    k, n, l = F[2].shape
    C = F[2].transpose(2, 0, 1).reshape(l * k, n)
    D = F[3].transpose(2, 0, 1).reshape(l * k, n)
    S, T = C + D, C - D
    return S @ S.T, T @ T.T


@pytest.mark.parametrize("kind", ["g_function", "linear", "polynomial"])
@pytest.mark.parametrize("k,n", [(6, 4096), (24, 1000), (32, 4096), (48, 4096), (64, 777), (96, 2048), (128, 4096)])
def test_fused_both_orders_against_the_oracle(ctx, kind, k, n):
    """vs_eval_reduce_first_second_order with NO device-produced value on the expected side: the
    Gram sums against S S^T / T T^T of the oracle's evaluation (orc.evaluate_scan; the one-kernel call folds its
    columns in its own segment shape, a few ulp per value from the oracle's -- far inside the bars), the record
    against the exactly rounded sums of the same values.  Replaces varsens/saltelli.py:329-355, 577-619."""
    from varsens_amd import _native
    P = params_for(kind, k, 1, 5 * k + n)
    d = design(ctx, k, n)
    obj = ctx.objective_create(KINDS[kind], k, 1, P)
    assert ctx.eval_reduce_both_supported(d, obj)
    nacc, ng = _native.acc_len(k, 1, 1), _native.gram_len(k)
    acc, g = ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
    ctx.eval_reduce_first_second_order(d, obj, 0, n, acc, g)
    S = orc.build_sample(k, n)
    F = orc.evaluate_scan(kind, P, S[0], S[1], fused=True)
    check_record(ctx.download(acc, (nacc,)), F, k, 1, rtol=1e-12)
    G = ctx.download(g, (2, k, k))
    SS, TT = gram_expected(F)
    np.testing.assert_allclose(G[0], SS, rtol=1e-11, atol=1e-12 * np.abs(SS).max())
    np.testing.assert_allclose(G[1], TT, rtol=1e-11, atol=1e-12 * np.abs(SS).max())


def fused_both_takes(k, l):
    """What include/varsens_hip.h promises for several outputs of a built-in separable objective (identity / linear
    scaling): 2 .. 8 outputs up to 32 index rows, k >= 16 up to 64, 2 or 4 outputs up to 128 (3 up to 112)."""
    K = k * l
    if l == 1:
        return k <= 128
    return (K <= 32 and l <= 8) or (K <= 64 and k >= 16) or (64 < K <= 128 and (l in (2, 4) or (l == 3 and K <= 112)))


@pytest.mark.parametrize("kind", ["g_function", "linear", "polynomial"])
@pytest.mark.parametrize("k,n,l", [(6, 1024, 2), (6, 4096, 3), (3, 700, 5), (2, 500, 8), (8, 64, 2),      # <= 16 index rows
                                   (16, 1000, 2), (4, 777, 7), (10, 3000, 3), (5, 1, 6),                   # <= 32
                                   (32, 4096, 2), (21, 1500, 3), (12, 2048, 5), (8, 129, 8), (17, 640, 2), (16, 3000, 4),
                                   (16, 777, 3), (15, 500, 3),                                             # <= 64
                                   (64, 2048, 2), (32, 3000, 4), (40, 1000, 2), (17, 5000, 4), (48, 300, 2),
                                   (25, 4096, 4), (56, 700, 2), (64, 1, 2), (30, 3000, 3), (37, 900, 3), (22, 4096, 3),
                                   (38, 500, 3)])                                                          # <= 128
def test_fused_both_orders_several_outputs_against_the_oracle(ctx, kind, k, n, l):
    """The one-kernel default call with SEVERAL outputs (the reference's two-output known-answer test,
    varsens/tests/test_g_function.py:77-88; sens_2[:, o, :, o'] slices of varsens/examples/varsens_earm_scipy.py:179-186):
    record and Gram sums (index row = column * l + output) against the oracle's evaluation, no device value on the
    expected side; and against the two-kernel route.  Replaces varsens/saltelli.py:329-355, 577-619."""
    from varsens_amd import _native
    P = params_for(kind, k, l, 3 * k + n + l)
    d = design(ctx, k, n)
    obj = ctx.objective_create(KINDS[kind], k, l, P)
    assert ctx.eval_reduce_both_supported(d, obj) == fused_both_takes(k, l)
    K = k * l
    nacc, ng = _native.acc_len(k, l, 1), _native.gram_len(K)
    acc, g = ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
    if not fused_both_takes(k, l):    # shapes the engine leaves to the two-kernel route (measured slower in one kernel)
        with pytest.raises(Exception, match="fused both-orders path covers"):
            ctx.eval_reduce_first_second_order(d, obj, 0, n, acc, g)
        return
    ctx.eval_reduce_first_second_order(d, obj, 0, n, acc, g)
    S = orc.build_sample(k, n)
    F = orc.evaluate_scan(kind, P, S[0], S[1], fused=True)
    rec = ctx.download(acc, (nacc,))
    check_record(rec, F, k, l, rtol=1e-12)
    G = ctx.download(g, (2, K, K))
    SS, TT = gram_expected_rows(F)
    np.testing.assert_allclose(G[0], SS, rtol=1e-11, atol=1e-12 * np.abs(SS).max())
    np.testing.assert_allclose(G[1], TT, rtol=1e-11, atol=1e-12 * np.abs(SS).max())
    assert np.array_equal(G[0], G[0].T) and np.array_equal(G[1], G[1].T)
    # the two-kernel route on the device's own f-values
    f = ctx.alloc(8 * 2 * (k + 1) * n * l)
    ctx.eval(d, obj, 0, n, f)
    acc1, g1 = ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
    ctx.reduce_first_second_order(f, n, k, l, acc1, g1)
    r1, G1 = ctx.download(acc1, (nacc,)), ctx.download(g1, (2, K, K))
    per = _native.acc_per_output(k)
    np.testing.assert_allclose(rec[4:4 + l * per], r1[4:4 + l * per], rtol=1e-12)
    np.testing.assert_allclose(G, G1, rtol=1e-12, atol=1e-13 * np.abs(G1).max())
    # bit-reproducible from call to call
    acc2, g2 = ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
    ctx.eval_reduce_first_second_order(d, obj, 0, n, acc2, g2)
    assert np.array_equal(ctx.download(acc2, (nacc,)), rec) and np.array_equal(ctx.download(g2, (2, K, K)), G)


def gram_expected_rows(F):
    """S S^T and T T^T with index row m = column * l + output (the layout of sens_2[k, l, k, l] flattened)."""
    k, n, l = F[2].shape
    C = F[2].transpose(0, 2, 1).reshape(k * l, n)
    D = F[3].transpose(0, 2, 1).reshape(k * l, n)
    S, T = C + D, C - D
    return S @ S.T, T @ T.T


@pytest.mark.parametrize("k,l,mode,generator", [(6, 2, 1, None), (20, 3, 1, None), (30, 4, 1, None), (6, 2, 0, "sobol"),
                                                (40, 2, 1, "sobol"), (5, 3, 2, None), (33, 2, 2, None), (16, 4, 2, None),
                                                (32, 4, 2, None), (37, 3, 2, "sobol")])
def test_fused_both_orders_several_outputs_designs(ctx, k, l, mode, generator):
    """Linear and power scaling and the Sobol generator take the several-output kernels too (the general column code;
    power scaling: round 4's last addition -- scale.magnitude of the reference's model studies reduces to it)."""
    from varsens_amd import _native
    from varsens_amd.saltelli import sobol_directions
    n = 2500
    lo = hi = None
    if mode == 1:
        lo, hi = np.linspace(-1, 0.5, k), np.linspace(1, 3, k)
    elif mode == 2:
        lo, hi = np.full(k, 1e-2), np.linspace(0.5, 1.0, k)
    d = design(ctx, k, n, mode, lo, hi, generator=generator, dirs=sobol_directions(k) if generator else None)
    P = params_for("g_function", k, l, k + l)
    obj = ctx.objective_create(KINDS["g_function"], k, l, P)
    K = k * l
    nacc, ng = _native.acc_len(k, l, 1), _native.gram_len(K)
    takes = not (mode == 2 and 32 < k * l <= 64)      # power scaling at 33 .. 64 index rows: measured slower, not routed
    assert ctx.eval_reduce_both_supported(d, obj) == takes
    if not takes:
        return
    f = ctx.alloc(8 * 2 * (k + 1) * n * l)
    ctx.eval(d, obj, 0, n, f)
    acc1, g1 = ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
    ctx.reduce_first_second_order(f, n, k, l, acc1, g1)
    acc, g = ctx.alloc(8 * nacc), ctx.alloc(8 * ng)
    ctx.eval_reduce_first_second_order(d, obj, 0, n, acc, g)
    per = _native.acc_per_output(k)
    r, r1 = ctx.download(acc, (nacc,)), ctx.download(acc1, (nacc,))
    np.testing.assert_allclose(r[4:4 + l * per], r1[4:4 + l * per], rtol=1e-12)
    np.testing.assert_allclose(r[4 + l * per:], r1[4 + l * per:], rtol=1e-9, atol=1e-9 * np.abs(r1[4 + l * per:]).max())
    G, G1 = ctx.download(g, (2, K, K)), ctx.download(g1, (2, K, K))
    np.testing.assert_allclose(G, G1, rtol=1e-12, atol=1e-13 * np.abs(G1).max())
