"""BASELINE.json configs at their full sizes on one MI355X.

At these sizes the oracle cannot evaluate everything in seconds, so parity is shown through
(a) spot rows: the device f-values of a few thousand rows (head, tail, random) against the oracle's
    values for exactly those rows (bit-exact for the algebraic objectives),
(b) size-independent properties of the reduction: power-of-two scaling is exact, shards add up to
    the whole, repeated runs are bit-identical,
(c) the analytic sensitivity indices within Monte Carlo tolerance.
"""
import numpy as np
import pytest

from oracle import saltelli_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from varsens_amd import _native
    c = _native.Context(0)
    yield c
    c.close()


def oracle_points(k, n, rows, scaling=None, first=None):
    """M_1[rows], M_2[rows] of the design (Halton, seed-1 shuffle) without building the rest."""
    first = orc.halton_skip(k) if first is None else first
    perm = orc.seed1_permutation(n)
    f = scaling if scaling is not None else (lambda p: p)
    M1 = np.concatenate([orc.halton(k, first + int(r), 1) for r in rows], axis=0)
    M2 = np.concatenate([orc.halton(k, first + n + int(perm[r]), 1) for r in rows], axis=0)
    return f(M1), f(M2)


def oracle_rows(k, n, rows, rows_func, scaling=None, first=None):
    """Oracle f-values [2k+2][len(rows)][l] for the given design rows only (rows_func: any objective)."""
    M1, M2 = oracle_points(k, n, rows, scaling, first)
    F = orc.evaluate_rows(rows_func, M1, M2, orc.column_swap(M2, M1), orc.column_swap(M1, M2))
    return np.concatenate([F[0][None], F[1][None], F[2], F[3]], axis=0)


def oracle_rows_separable(kind, params, k, n, rows, strict=False):
    """The same for a built-in separable objective in the engine's order (scan for k <= 256, or strict)."""
    M1, M2 = oracle_points(k, n, rows)
    F = orc.evaluate_engine_order(kind, params, M1, M2, strict=strict)
    return np.concatenate([F[0][None], F[1][None], F[2], F[3]], axis=0)


def device_eval(ctx, k, n, kind, params, l=1, mode=0, lo=None, hi=None, begin=0, count=None):
    from varsens_amd import _native
    count = n if count is None else count
    perm = orc.seed1_permutation(n).astype(np.uint32)
    d = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), mode, lo, hi, perm)
    o = ctx.objective_create(kind, k, l, params)
    f = ctx.alloc(8 * 2 * (k + 1) * count * l)
    ctx.eval(d, o, begin, count, f)
    ctx.sync()
    return f


def spot(ctx, f, k, rows_total, l, rows):
    out = np.empty((2 * k + 2, len(rows), l))
    for b in range(2 * k + 2):
        for i, r in enumerate(rows):
            for o in range(l):
                # row-blocked f-values (include/varsens_hip.h): row block R of 2^s rows, then plane o (2k+2) + b
                step = 1 << ctx.f_shift
                first = (int(r) // step) * step
                pitch = min(step, rows_total - first)
                at = first * l * (2 * k + 2) + (o * (2 * k + 2) + b) * pitch + (int(r) - first)
                out[b, i, o] = ctx.download(f, (1,), byte_offset=8 * at)[0]
    return out


def first_order(ctx, f, rows, n, k, l=1):
    from varsens_amd import _native
    acc = ctx.alloc(8 * _native.acc_len(k, l, 1))
    ctx.reduce_first_order(f, rows, k, l, acc)
    out = ctx.alloc(8 * _native.first_order_len(k, l))
    ctx.finish_first_order(acc, n, k, l, out)
    return ctx.download(acc, (_native.acc_len(k, l, 1),)), ctx.download(out, (2 + 4 * k, l))


def pick_rows(n, seed, m=48):
    rng = np.random.RandomState(seed)
    return np.unique(np.concatenate([np.arange(3), np.arange(n - 3, n), rng.randint(0, n, m)]))


def test_config2_gfunction_k6_n2pow20(ctx):
    from varsens_amd import _native
    k, n = 6, 1 << 20
    a = [0, 0.5, 3, 9, 99, 99]
    f = device_eval(ctx, k, n, _native.OBJ_G_FUNCTION, a)
    rows = pick_rows(n, 2)
    got = spot(ctx, f, k, n, 1, rows)
    assert np.array_equal(got, oracle_rows_separable("g_function", a, k, n, rows))
    strict = oracle_rows_separable("g_function", a, k, n, rows, strict=True)
    assert np.max(np.abs(got - strict) / np.abs(strict)) <= 1e-15
    rec, fin = first_order(ctx, f, n, n, k)
    V = orc.g_partial_variances(a)
    total, S, ST = orc.product_form_indices(V)
    sens, sens_t = fin[2 + 2 * k:2 + 3 * k, 0], fin[2 + 3 * k:2 + 4 * k, 0]
    assert abs(fin[1, 0] - total) < 2e-3 and abs(fin[0, 0] - 1.0) < 1e-3
    # quasi-Monte Carlo error at n = 2^20 (reference's own profile: rms ~1e-4 at this n)
    np.testing.assert_allclose(sens, S, atol=2e-3)
    np.testing.assert_allclose(sens_t, ST, atol=2e-3)
    # reproducible to the bit
    rec2, fin2 = first_order(ctx, f, n, n, k)
    assert np.array_equal(rec, rec2) and np.array_equal(fin, fin2)


def test_config2_against_the_reference_run(ctx, golden, capsys):
    """BASELINE configs[1] against fixture G15 = the reference's own 3-step run at k=6, n=2^20 (14.7 M calls
    of its Python objective, tests/golden/make_golden.py --config2).  The device f-values ARE the
    reference's (same SHA-256 over all 14 680 064 of them); raw sums are held to exactly rounded sums,
    the indices to their evaluation from those sums (north-star bar 1e-10), and the reference's own
    distance to that evaluation -- its summation noise, SURVEY.md section 7 -- is printed beside it."""
    import hashlib
    from varsens_amd import _native
    g = golden("g15_gfunc_k6_n1048576")
    k, n = 6, 1 << 20
    a = [0, 0.5, 3, 9, 99, 99]
    ctx.set_eval_order("strict")       # numpy.prod's left-to-right product, as the reference's objective multiplies
    try:
        f = device_eval(ctx, k, n, _native.OBJ_G_FUNCTION, a)
    finally:
        ctx.set_eval_order("fast")
    blocks = ctx.download_blocks(f, k, 1, n)          # [2k+2][n][1], flat order
    assert hashlib.sha256(np.ascontiguousarray(blocks.reshape(-1, 1)).tobytes()).hexdigest() == str(g["fvals_sha"])
    assert np.array_equal(blocks[0][:8], g["fM_1_head"]) and np.array_equal(blocks[2:2 + k, :4], g["fN_j_head"])
    rec, fin = first_order(ctx, f, n, n, k)
    F = (blocks[0], blocks[1], blocks[2:2 + k], blocks[2 + k:])
    ex = orc.exact_accumulators(*F)
    per = _native.acc_per_output(k)
    body = rec[4:4 + per]
    np.testing.assert_allclose(body[0], ex["sab"][0], rtol=1e-13)
    for q, name in enumerate(("ac", "bd", "ad", "bc")):
        np.testing.assert_allclose(body[3 + q * k:3 + (q + 1) * k], ex[name][:, 0], rtol=1e-13, err_msg=name)
    want = orc.indices_from_accumulators(ex, n)
    for row, name in ((0, "E_2"), (1, "var_y")):
        np.testing.assert_allclose(fin[row, 0], want[name][0], rtol=1e-13, err_msg=name)
        np.testing.assert_allclose(fin[row, 0], g[name][0], rtol=1e-13, err_msg=name)     # the reference's value
    sens, sens_t = fin[2 + 2 * k:2 + 3 * k, 0], fin[2 + 3 * k:2 + 4 * k, 0]
    np.testing.assert_allclose(sens, want["sens"][:, 0], rtol=1e-10, atol=0)
    np.testing.assert_allclose(sens_t, want["sens_t"][:, 0], rtol=1e-10, atol=0)
    # the reference against the same exact evaluation: its own rounding noise (S_5, S_6 are small differences)
    ref_dist = np.abs(g["sens"][:, 0] - want["sens"][:, 0]) / np.abs(want["sens"][:, 0])
    gpu_dist = np.abs(sens - want["sens"][:, 0]) / np.abs(want["sens"][:, 0])
    with capsys.disabled():
        print("\nconfig 2, relative distance of sens to the exactly summed evaluation: reference %s | GPU %s"
              % (np.array2string(ref_dist, precision=1), np.array2string(gpu_dist, precision=1)))
    assert gpu_dist.max() <= max(ref_dist.max(), 1e-12)
    np.testing.assert_allclose(sens, g["sens"][:, 0], rtol=2e-9)              # within the reference's own noise
    np.testing.assert_allclose(sens_t, g["sens_t"][:, 0], rtol=2e-9, atol=1e-12)
    # second order against the reference's tensordot results
    gram = ctx.alloc(8 * 2 * k * k)
    ctx.reduce_second_order(f, n, k, 1, gram)
    s2, s2n = ctx.alloc(8 * k * k), ctx.alloc(8 * k * k)
    ctx.finish_second_order(gram, ctx.upload(fin), n, k, 1, s2, s2n)
    np.testing.assert_allclose(ctx.download(s2, (k, 1, k, 1)), g["sens_2"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ctx.download(s2n, (k, 1, k, 1)), g["sens_2n"], rtol=1e-9, atol=1e-12)
    # the default (fast) evaluation order: f-values within 1 ulp-level of the reference's, same indices
    f2 = device_eval(ctx, k, n, _native.OBJ_G_FUNCTION, a)
    fast = ctx.download_blocks(f2, k, 1, n)
    assert np.max(np.abs(fast - blocks) / np.abs(blocks)) <= 1e-15
    rec2, fin2 = first_order(ctx, f2, n, n, k)
    np.testing.assert_allclose(fin2[:2 + 2 * k], fin[:2 + 2 * k], rtol=1e-13)
    np.testing.assert_allclose(fin2[2 + 2 * k:, 0], fin[2 + 2 * k:, 0], rtol=1e-10)
    # ... and the whole of configs[1] in ONE kernel (vs_eval_reduce_first_second_order, no f-values in HBM): raw sums and
    # indices held to the same exactly summed evaluation and to the reference's own outputs
    d = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, orc.seed1_permutation(n).astype(np.uint32))
    o = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, np.asarray(a, dtype=float))
    assert ctx.eval_reduce_both_supported(d, o)
    nacc = _native.acc_len(k, 1, 1)
    acc3, gram3, first3 = ctx.alloc(8 * nacc), ctx.alloc(8 * 2 * k * k), ctx.alloc(8 * _native.first_order_len(k, 1))
    ctx.eval_reduce_first_second_order(d, o, 0, n, acc3, gram3)
    ctx.finish_first_order(acc3, n, k, 1, first3)
    rec3, fin3 = ctx.download(acc3, (nacc,)), ctx.download(first3, (2 + 4 * k, 1))
    body3 = rec3[4:4 + per]
    for q, name in enumerate(("ac", "bd", "ad", "bc")):
        np.testing.assert_allclose(body3[3 + q * k:3 + (q + 1) * k], ex[name][:, 0], rtol=1e-12, err_msg="fused " + name)
    np.testing.assert_allclose(fin3[2 + 2 * k:2 + 3 * k, 0], want["sens"][:, 0], rtol=1e-10, err_msg="fused sens")
    np.testing.assert_allclose(fin3[2 + 3 * k:2 + 4 * k, 0], want["sens_t"][:, 0], rtol=1e-10, err_msg="fused sens_t")
    np.testing.assert_allclose(fin3[2 + 2 * k:2 + 3 * k, 0], g["sens"][:, 0], rtol=2e-9)
    ctx.finish_second_order(gram3, first3, n, k, 1, s2, s2n)
    np.testing.assert_allclose(ctx.download(s2, (k, 1, k, 1)), g["sens_2"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ctx.download(s2n, (k, 1, k, 1)), g["sens_2n"], rtol=1e-9, atol=1e-12)


def test_config3_ishigami_k3_n2pow22(ctx):
    from varsens_amd import _native
    k, n = 3, 1 << 22
    lo, hi = np.full(3, -np.pi), np.full(3, np.pi)
    f = device_eval(ctx, k, n, _native.OBJ_ISHIGAMI, [7.0, 0.1], mode=1, lo=lo, hi=hi)
    rows = pick_rows(n, 3)
    got = spot(ctx, f, k, n, 1, rows)
    want = oracle_rows(k, n, rows, orc.ishigami_rows, scaling=lambda x: orc.scale_linear(x, lo, hi))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)      # device sin vs libm sin
    rec, fin = first_order(ctx, f, n, n, k)
    V, S, ST = orc.ishigami_truth()
    np.testing.assert_allclose(fin[1, 0], V, rtol=2e-3)
    np.testing.assert_allclose(fin[2 + 2 * k:2 + 3 * k, 0], S, atol=3e-3)
    np.testing.assert_allclose(fin[2 + 3 * k:2 + 4 * k, 0], ST, atol=3e-3)
    # the non-separable pair (x1,x3): closed second-order index V_1 + V_3 + V_13 (sens_2 cross term)
    gram = ctx.alloc(8 * 2 * k * k)
    ctx.reduce_second_order(f, n, k, 1, gram)
    from varsens_amd import _native as nat
    first = ctx.upload(fin)
    s2, s2n = ctx.alloc(8 * k * k), ctx.alloc(8 * k * k)
    ctx.finish_second_order(gram, first, n, k, 1, s2, s2n)
    sens_2 = ctx.download(s2, (k, k))
    V1, V3, V13 = 0.5 * (1 + 0.1 * np.pi ** 4 / 5) ** 2, 0.0, 8 * 0.01 * np.pi ** 8 / 225
    np.testing.assert_allclose(sens_2[0, 2], (V1 + V3 + V13) / V, atol=5e-3)
    np.testing.assert_allclose(sens_2[0, 2], sens_2[2, 0], rtol=1e-12)
    # configs[2] in one kernel (the default call's route): same record and Gram sums up to summation order
    d = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 1, lo, hi, orc.seed1_permutation(n).astype(np.uint32))
    o = ctx.objective_create(_native.OBJ_ISHIGAMI, k, 1, np.array([7.0, 0.1]))
    nacc = _native.acc_len(k, 1, 1)
    acc3, gram3 = ctx.alloc(8 * nacc), ctx.alloc(8 * 2 * k * k)
    ctx.eval_reduce_first_second_order(d, o, 0, n, acc3, gram3)
    per = _native.acc_per_output(k)
    np.testing.assert_allclose(ctx.download(acc3, (nacc,))[4:4 + per], rec[4:4 + per], rtol=1e-11)
    G1 = ctx.download(gram, (2, k, k))
    np.testing.assert_allclose(ctx.download(gram3, (2, k, k)), G1, rtol=1e-11, atol=1e-12 * np.abs(G1).max())


def test_config4_linear_k32_n2pow22_properties(ctx):
    from varsens_amd import _native
    k, n = 32, 1 << 22
    c = np.arange(1, k + 1, dtype=float)
    f = device_eval(ctx, k, n, _native.OBJ_LINEAR, c)
    rows = pick_rows(n, 4, 24)
    assert np.array_equal(spot(ctx, f, k, n, 1, rows), oracle_rows_separable("linear", c, k, n, rows))
    rec, fin = first_order(ctx, f, n, n, k)
    truth = c * c / np.sum(c * c)
    np.testing.assert_allclose(fin[2 + 2 * k:2 + 3 * k, 0], truth, atol=5e-4)
    np.testing.assert_allclose(fin[2 + 3 * k:2 + 4 * k, 0], truth, atol=5e-4)
    np.testing.assert_allclose(fin[1, 0], np.sum(c * c) / 12.0, rtol=1e-3)
    np.testing.assert_allclose(fin[0, 0], (np.sum(c) / 2.0) ** 2, rtol=1e-4)
    # shards add up to the whole: 8 row shards (the multi-GPU split) summed on the host, then finished
    from varsens_amd.dist import shard_rows
    world = 8
    nacc = _native.acc_len(k, 1, world)
    total = np.zeros(nacc)
    for rank in range(world):
        b, cnt = shard_rows(n, world, rank)
        fs = device_eval(ctx, k, n, _native.OBJ_LINEAR, c, begin=b, count=cnt)
        acc = ctx.alloc(8 * nacc)
        ctx.reduce_first_order(fs, cnt, k, 1, acc, 0, rank, world)
        total += ctx.download(acc, (nacc,))
        fs.free()
    out = ctx.alloc(8 * _native.first_order_len(k, 1))
    ctx.finish_first_order(ctx.upload(total), n, k, 1, out, world)
    fin8 = ctx.download(out, (2 + 4 * k, 1))
    np.testing.assert_allclose(fin8[:2 + 2 * k], fin[:2 + 2 * k], rtol=1e-13)
    np.testing.assert_allclose(fin8[2 + 2 * k:], fin[2 + 2 * k:], rtol=1e-10, atol=1e-14)


def test_reduction_power_of_two_scaling_is_exact(ctx):
    """Doubling every f-value multiplies every sum of products by exactly 4 (no data-dependent path)."""
    from varsens_amd import _native
    rng = np.random.RandomState(8)
    k, rows = 16, 1 << 18
    blocks = rng.randn(2 * k + 2, rows, 1)
    r1, f1 = first_order(ctx, ctx.upload_blocks(blocks), rows, rows, k)
    r2, f2 = first_order(ctx, ctx.upload_blocks(2.0 * blocks), rows, rows, k)
    per = _native.acc_per_output(k)
    body1, body2 = r1[4:4 + per], r2[4:4 + per]
    assert np.array_equal(body2[0], 4.0 * body1[0]) and np.array_equal(body2[3:], 4.0 * body1[3:])
    assert np.array_equal(body2[1:3], 2.0 * body1[1:3])
    assert np.array_equal(f2[2 + 2 * k:], f1[2 + 2 * k:])      # sens, sens_t are scale-invariant, to the bit


def test_config5_shard_polynomial_k128(ctx):
    """configs[4]: k=128, n=2^24 over 8 GPUs -- here rank 3's shard (2^21 rows, 4.3 GB of f-values)."""
    from varsens_amd import _native
    from varsens_amd.dist import shard_rows
    k, n = 128, 1 << 24
    beta = 1.0 / (1.0 + np.arange(k) / 8.0)
    begin, count = shard_rows(n, 8, 3)
    assert count == 1 << 21
    f = device_eval(ctx, k, n, _native.OBJ_POLYNOMIAL, beta, begin=begin, count=count)
    local = np.array([0, 1, count // 2, count - 1])
    got = spot(ctx, f, k, count, 1, local)
    assert np.array_equal(got, oracle_rows_separable("polynomial", beta, k, n, begin + local))
    rec, fin = first_order(ctx, f, count, count, k)       # the shard alone is a valid (smaller) design
    V = beta ** 2 / 12.0
    total, S, ST = orc.product_form_indices(V)
    np.testing.assert_allclose(fin[1, 0], total, rtol=2e-2)
    np.testing.assert_allclose(fin[2 + 2 * k:2 + 3 * k, 0], S, atol=5e-3)


def test_config5_whole_polynomial_k128_n2pow24(ctx):
    """configs[4] whole on ONE MI355X: k=128, n=2^24, 34.6 GB of f-values -- evaluation (scan kernels, k >= 33),
    first order, second order.  Spot rows against the oracle (bit-exact in the kernels' association order,
    rounding-close to the strict left-to-right product), analytic S_i / ST_i, and the 8-rank row split:
    shard records and shard Gram sums add up to the whole."""
    from varsens_amd import _native
    from varsens_amd.dist import shard_rows
    k, n = 128, 1 << 24
    beta = 1.0 / (1.0 + np.arange(k) / 8.0)
    f = device_eval(ctx, k, n, _native.OBJ_POLYNOMIAL, beta)
    rows = pick_rows(n, 5, 26)
    got = spot(ctx, f, k, n, 1, rows)
    assert np.array_equal(got, oracle_rows_separable("polynomial", beta, k, n, rows))
    strict = oracle_rows_separable("polynomial", beta, k, n, rows, strict=True)
    assert np.max(np.abs(got - strict) / np.abs(strict)) <= 6e-15
    rec, fin = first_order(ctx, f, n, n, k)
    V = beta ** 2 / 12.0
    total, S, ST = orc.product_form_indices(V)
    # quasi-Monte Carlo error of a 128-dimensional Halton design at n = 2^24: var_y 0.932151 vs 0.934723 analytic
    # (2.8e-3; the same digits as round 1's chain-kernel run, profiles/r01/config5_k128_n2pow24_single_gpu.log)
    np.testing.assert_allclose(fin[1, 0], total, rtol=5e-3)
    np.testing.assert_allclose(fin[0, 0], 1.0, rtol=1e-3)
    np.testing.assert_allclose(fin[2 + 2 * k:2 + 3 * k, 0], S, atol=1e-3)      # builder's full-size run: 3e-4 / 8e-4
    np.testing.assert_allclose(fin[2 + 3 * k:2 + 4 * k, 0], ST, atol=2e-3)
    # second order on the whole: closed pair indices (V_i + V_j + V_i V_j) / V of a product form
    gram = ctx.alloc(8 * 2 * k * k)
    ctx.reduce_second_order(f, n, k, 1, gram)
    G = ctx.download(gram, (2, k, k))
    s2, s2n = ctx.alloc(8 * k * k), ctx.alloc(8 * k * k)
    ctx.finish_second_order(gram, ctx.upload(fin), n, k, 1, s2, s2n)
    sens_2 = ctx.download(s2, (k, k))
    pair = (V[:, None] + V[None, :] + V[:, None] * V[None, :]) / total
    off = ~np.eye(k, dtype=bool)
    np.testing.assert_allclose(sens_2[off], pair[off], atol=3e-3)
    np.testing.assert_allclose(sens_2, sens_2.T, rtol=1e-12, atol=1e-15)
    f.free()
    # the 8-rank row split (what the 8-GPU run does, one shard at a time here): records and Gram sums add up
    world = 8
    nacc = _native.acc_len(k, 1, world)
    total_rec, total_gram = np.zeros(nacc), np.zeros((2, k, k))
    for rank in range(world):
        b, cnt = shard_rows(n, world, rank)
        fs = device_eval(ctx, k, n, _native.OBJ_POLYNOMIAL, beta, begin=b, count=cnt)
        acc, gs = ctx.alloc(8 * nacc), ctx.alloc(8 * 2 * k * k)
        ctx.reduce_first_second_order(fs, cnt, k, 1, acc, gs, 0, rank, world)
        total_rec += ctx.download(acc, (nacc,))
        total_gram += ctx.download(gs, (2, k, k))
        fs.free()
    out = ctx.alloc(8 * _native.first_order_len(k, 1))
    ctx.finish_first_order(ctx.upload(total_rec), n, k, 1, out, world)
    fin8 = ctx.download(out, (2 + 4 * k, 1))
    np.testing.assert_allclose(fin8[:2 + 2 * k], fin[:2 + 2 * k], rtol=1e-13)
    np.testing.assert_allclose(fin8[2 + 2 * k:], fin[2 + 2 * k:], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(total_gram, G, rtol=1e-12, atol=1e-12 * np.abs(G).max())
    # configs[4] without its 34.6 GB of f-values: the one-kernel call (eight waves x 16 columns, 137 KiB tile), whole and
    # as the 8-rank split -- records and Gram sums equal the two-kernel path's up to summation order
    d = ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), 0, None, None, orc.seed1_permutation(n).astype(np.uint32))
    o = ctx.objective_create(_native.OBJ_POLYNOMIAL, k, 1, beta)
    assert ctx.eval_reduce_both_supported(d, o)
    per = _native.acc_per_output(k)
    acc1, g1 = ctx.alloc(8 * _native.acc_len(k, 1, 1)), ctx.alloc(8 * 2 * k * k)
    ctx.eval_reduce_first_second_order(d, o, 0, n, acc1, g1)
    np.testing.assert_allclose(ctx.download(acc1, (_native.acc_len(k, 1, 1),))[4:4 + per], rec[4:4 + per], rtol=1e-11)
    np.testing.assert_allclose(ctx.download(g1, (2, k, k)), G, rtol=1e-11, atol=1e-12 * np.abs(G).max())
    fused_rec, fused_gram = np.zeros(nacc), np.zeros((2, k, k))
    for rank in range(world):
        b, cnt = shard_rows(n, world, rank)
        acc, gs = ctx.alloc(8 * nacc), ctx.alloc(8 * 2 * k * k)
        ctx.eval_reduce_first_second_order(d, o, b, cnt, acc, gs, rank, world)
        fused_rec += ctx.download(acc, (nacc,))
        fused_gram += ctx.download(gs, (2, k, k))
    ctx.finish_first_order(ctx.upload(fused_rec), n, k, 1, out, world)
    finf = ctx.download(out, (2 + 4 * k, 1))
    np.testing.assert_allclose(finf[:2 + 2 * k], fin[:2 + 2 * k], rtol=1e-11)
    np.testing.assert_allclose(finf[2 + 2 * k:], fin[2 + 2 * k:], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(fused_gram, G, rtol=1e-11, atol=1e-12 * np.abs(G).max())


def test_model_study_shape_k100_l10():
    """A shape like the reference's model studies: k=100 parameters, l=10 outputs, n=20000 draws, a user
    objective compiled by hiprtc.  First and second order (k*l = 1000 index rows, 528 super-block pairs)
    against the oracle's numpy arithmetic on the same f-values."""
    import varsens_amd as vs
    k, l, n = 100, 10, 20000
    body = """
    for (int o = 0; o < %d; ++o) {
        double s = 0.0;
        for (int c = 0; c < k; ++c) s += p[(o * 7 + c) %% k] * x[c] * (1.0 + 0.01 * o * x[(c + o + 1) %% k]);
        out[o] = s;
    }""" % l
    obj = vs.objectives.Expression(body, outputs=l, params=1.0 / (1.0 + np.arange(k)))
    s = vs.Sample(k, n, vs.scale.Linear(-1.0, 2.0), verbose=False)
    o = vs.Objective(k, n, s, obj, verbose=False)
    v = vs.Varsens(o, verbose=False)
    R = orc.compute_varsens(o.fM_1, o.fM_2, o.fN_j, o.fN_nj, n)
    assert v.sens_2.shape == (k, l, k, l)
    for name in ("E_2", "var_y", "U_j", "U_nj"):
        np.testing.assert_allclose(getattr(v, name), R[name], rtol=1e-12, err_msg=name)
    for name in ("sens", "sens_t", "sens_2", "sens_2n"):
        a, b = getattr(v, name), R[name]
        assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), name
