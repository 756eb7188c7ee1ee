"""The row-blocked f-value layout (include/varsens_hip.h "F-VALUE LAYOUT", csrc/vs_device_gen.h: vs_f_tile).

Every kernel that writes or reads objective values works on row tiles inside one row block; which rows share a row
block must not change a single bit of any result.  Each case below runs under three contexts -- row blocks of 2^11 rows
(VS_F_SHIFT=11, so that a few thousand rows already span several row blocks and a ragged last one), the default 2^12, and
2^40 (one row block: the planar layout of ABI <= 7) -- and asks for (a) the evaluation kernels' values in the layout
numpy builds on the host from the oracle's values (bit for bit), (b) identical bits of every reduction across the three
layouts, (c) the sums of one of them against exactly rounded sums of the inputs."""
import math

import numpy as np
import pytest

from oracle import saltelli_oracle as orc

pytestmark = pytest.mark.gpu

SHIFTS = (11, 12, 40)


@pytest.fixture(scope="module")
def contexts():
    import os
    from varsens_amd import _native
    made = {}
    saved = {name: os.environ.get(name) for name in ("VS_TUNING", "VS_F_SHIFT")}
    try:
        for sh in SHIFTS:
            os.environ["VS_TUNING"] = "1"
            os.environ["VS_F_SHIFT"] = str(sh)
            made[sh] = _native.Context(0)
            assert made[sh].f_shift == sh
    finally:
        for name, value in saved.items():
            if value is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = value
    yield made
    for c in made.values():
        c.close()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_host_statement_of_the_layout():
    """to_device_layout / from_device_layout (numpy): the formula of include/varsens_hip.h, element by element"""
    from varsens_amd import _native
    k, l, rows, sh = 2, 3, 37, 4          # row blocks of 16 rows: 16 + 16 + 5
    blocks = np.arange((2 * k + 2) * rows * l, dtype=float).reshape(2 * k + 2, rows, l)
    flat = _native.to_device_layout(blocks, sh)
    P = l * (2 * k + 2)
    assert flat.shape == (P * rows,)
    for b in range(2 * k + 2):
        for r in range(rows):
            for o in range(l):
                R = r >> sh
                first = R << sh
                pitch = min(1 << sh, rows - first)
                assert flat[first * P + (o * (2 * k + 2) + b) * pitch + (r - first)] == blocks[b, r, o]
    assert np.array_equal(_native.from_device_layout(flat, k, l, rows, sh), blocks)
    # one row block: plane after plane
    assert np.array_equal(_native.to_device_layout(blocks, 6), blocks.transpose(2, 0, 1).reshape(-1))


def _design(ctx, k, n, mode=0, lo=None, hi=None):
    from varsens_amd import _native
    return ctx.design_create(_native.GEN_HALTON, k, n, orc.halton_skip(k), mode, lo, hi,
                             orc.seed1_permutation(n).astype(np.uint32))


@pytest.mark.parametrize("kind,k,n,l", [("g_function", 6, 5000, 1), ("g_function", 6, 4200, 3), ("linear", 33, 4500, 1),
                                        ("polynomial", 70, 2049, 2), ("g_function", 130, 4097, 1)])
def test_scan_evaluation_lands_where_the_layout_says(contexts, kind, k, n, l):
    """vs_eval (scan kernels; several outputs through the MULTI walk): the device buffer, read raw, is the host's layout of
    the oracle's values -- under every row-block size"""
    from varsens_amd import _native
    kinds = {"g_function": _native.OBJ_G_FUNCTION, "linear": _native.OBJ_LINEAR, "polynomial": _native.OBJ_POLYNOMIAL}
    rng = np.random.RandomState(k * 7 + l)
    params = rng.uniform(0.0, 9.0, (l, k))
    S = orc.build_sample(k, n)
    F = orc.evaluate_engine_order(kind, params, S[0], S[1])
    want = np.concatenate([F[0][None], F[1][None], F[2], F[3]], axis=0)      # [2k+2][n][l]
    for sh, ctx in contexts.items():
        d = _design(ctx, k, n)
        obj = ctx.objective_create(kinds[kind], k, l, params)
        f = ctx.alloc(8 * 2 * (k + 1) * n * l)
        ctx.eval(d, obj, 0, n, f)
        raw = ctx.download(f, (2 * (k + 1) * n * l,))
        assert np.array_equal(_bits(raw), _bits(_native.to_device_layout(want, sh))), sh
        assert np.array_equal(_bits(ctx.download_values(f, n, k, l)), _bits(want.reshape(-1, l))), sh
        f.free()


def test_other_evaluation_kernels_land_there_too(contexts):
    """strict-order chain kernels, Ishigami, invert (l = k) and objectives compiled at run time"""
    from varsens_amd import _native
    for sh, ctx in contexts.items():
        # Ishigami, thread per row
        n = 4200
        lo, hi = np.full(3, -math.pi), np.full(3, math.pi)
        d = _design(ctx, 3, n, _native.SCALE_LINEAR, lo, hi)
        obj = ctx.objective_create(_native.OBJ_ISHIGAMI, 3, 1, [7.0, 0.1])
        f = ctx.alloc(8 * 8 * n)
        ctx.eval(d, obj, 0, n, f)
        got = ctx.download_blocks(f, 3, 1, n)
        S = orc.build_sample(3, n, lambda x: orc.scale_linear(x, lo, hi))
        F = orc.evaluate_rows(orc.ishigami_rows, *S)
        np.testing.assert_allclose(got, np.concatenate([F[0][None], F[1][None], F[2], F[3]], axis=0), rtol=0, atol=1e-14)
        assert np.array_equal(_bits(ctx.download(f, (8 * n,))), _bits(_native.to_device_layout(got, sh)))
        f.free()
        # invert: l = k outputs, general kernel
        k, n = 5, 4100
        d = _design(ctx, k, n)
        obj = ctx.objective_create(_native.OBJ_INVERT, k, k, None)
        f = ctx.alloc(8 * 2 * (k + 1) * n * k)
        ctx.eval(d, obj, 0, n, f)
        S = orc.build_sample(k, n)
        flat = orc.flat_sample(*S)
        assert np.array_equal(ctx.download_values(f, n, k, k), 1.0 - flat)
        assert np.array_equal(_bits(ctx.download(f, (2 * (k + 1) * n * k,))),
                              _bits(_native.to_device_layout((1.0 - flat).reshape(2 * k + 2, n, k), sh)))
        f.free()
        # strict left-to-right order: the chain kernels
        ctx.set_eval_order("strict")
        a = np.array([0.0, 0.5, 3.0, 9.0, 99.0])
        obj = ctx.objective_create(_native.OBJ_G_FUNCTION, k, 1, a)
        f = ctx.alloc(8 * 2 * (k + 1) * n)
        ctx.eval(d, obj, 0, n, f)
        ctx.set_eval_order("fast")
        F = orc.evaluate_engine_order("g_function", a, S[0], S[1], strict=True)
        want = np.concatenate([F[0][None], F[1][None], F[2], F[3]], axis=0)
        assert np.array_equal(_bits(ctx.download(f, (2 * (k + 1) * n,))), _bits(_native.to_device_layout(want, sh)))
        f.free()
        # run-time compiled: a general body (two outputs) and a separable term
        body = "out[0] = x[0] + 2.0 * x[1] * x[2]; out[1] = x[3] - x[4];"
        obj = ctx.objective_create_source(k, 2, body, None)
        f = ctx.alloc(8 * 2 * (k + 1) * n * 2)
        ctx.eval(d, obj, 0, n, f)
        want = np.stack([flat[:, 0] + 2.0 * flat[:, 1] * flat[:, 2], flat[:, 3] - flat[:, 4]], axis=1)
        assert np.array_equal(ctx.download_values(f, n, k, 2), want)
        f.free()
        obj = ctx.objective_create_separable(k, "return (fabs(4.0 * x - 2.0) + p[c]) / (1.0 + p[c]);", False, a)
        f = ctx.alloc(8 * 2 * (k + 1) * n)
        ctx.eval(d, obj, 0, n, f)
        F = orc.evaluate_scan("g_function", a, S[0], S[1])
        want = np.concatenate([F[0][None], F[1][None], F[2], F[3]], axis=0)
        got = ctx.download_values(f, n, k, 1).reshape(2 * k + 2, n, 1)
        np.testing.assert_allclose(got, want, rtol=1e-14)
        f.free()


def _exact_dot(x, y):
    return math.fsum((x * y).tolist())


# (k, rows, l, nan_rows): every second-order family and both first-order forms, rows across 2-3 row blocks of 2^11 with a
# ragged last one; odd row counts (8-byte loads); NaN-trimmed passes
CASES = [(6, 5000, 1, 0), (6, 4097, 1, 0), (20, 4500, 1, 0), (32, 6000, 1, 0), (40, 4200, 1, 0), (40, 4101, 1, 0),
         (56, 4100, 1, 0), (100, 4200, 1, 0), (100, 4099, 1, 0), (140, 2100, 1, 0), (10, 4200, 3, 0), (24, 2200, 3, 0),
         (6, 5000, 1, 3), (40, 4200, 1, 2), (100, 4200, 1, 2), (140, 2100, 1, 1), (5, 4300, 2, 2)]


@pytest.mark.parametrize("k,rows,l,nans", CASES)
def test_reductions_do_not_depend_on_the_row_block_size(contexts, k, rows, l, nans):
    from varsens_amd import _native
    rng = np.random.RandomState(k * 100 + rows % 97 + l)
    blocks = 1.0 + rng.rand(2 * k + 2, rows, l)
    flags = 0
    if nans:
        flags = _native.REDUCE_NAN_TRIM
        for i in range(nans):
            blocks[(3 * i) % (2 * k + 2), (2049 * (i + 1) + 5) % rows, 0] = np.nan
    nacc = _native.acc_len(k, l, 1)
    K = k * l
    out = {}
    for sh, ctx in contexts.items():
        f = ctx.upload_blocks(blocks)
        acc, acc2, g1, g2 = ctx.alloc(8 * nacc), ctx.alloc(8 * nacc), ctx.alloc(8 * 2 * K * K), ctx.alloc(8 * 2 * K * K)
        ctx.reduce_first_order(f, rows, k, l, acc, flags)
        ctx.reduce_second_order(f, rows, k, l, g1, flags)
        ctx.reduce_first_second_order(f, rows, k, l, acc2, g2, flags)
        out[sh] = [ctx.download(acc, (nacc,)), ctx.download(g1, (2, K, K)), ctx.download(acc2, (nacc,)),
                   ctx.download(g2, (2, K, K))]
        for b in (f, acc, acc2, g1, g2):
            b.free()
    for sh in SHIFTS[1:]:
        for a, b in zip(out[SHIFTS[0]], out[sh]):
            assert np.array_equal(_bits(a), _bits(b)), (sh, k, rows, l)
    # ... and they are the right sums (one output column against exactly rounded sums)
    keep = ~np.isnan(blocks[:, :, 0]).any(axis=0)
    rec, gram = out[11][0], out[11][1]
    assert rec[0] == keep.sum() and rec[1] == rows - keep.sum()
    a, b = blocks[0, keep, 0], blocks[1, keep, 0]
    per = _native.acc_per_output(k)
    r0 = rec[_native.ACC_HEADER:_native.ACC_HEADER + per]
    assert abs(r0[0] - _exact_dot(a, b)) <= 1e-13 * abs(r0[0])
    for j in (0, k // 2, k - 1):
        c, d = blocks[2 + j, keep, 0], blocks[2 + k + j, keep, 0]
        for got, want in ((r0[3 + j], _exact_dot(a, c)), (r0[3 + k + j], _exact_dot(b, d)),
                          (r0[3 + 2 * k + j], _exact_dot(a, d)), (r0[3 + 3 * k + j], _exact_dot(b, c))):
            assert abs(got - want) <= 1e-13 * abs(want)
    C = blocks[2:2 + k][:, keep].transpose(0, 2, 1).reshape(K, -1)      # index row = column * l + output
    D = blocks[2 + k:][:, keep].transpose(0, 2, 1).reshape(K, -1)
    if not np.isnan(C).any() and not np.isnan(D).any():
        np.testing.assert_allclose(gram[0], (C + D) @ (C + D).T, rtol=1e-11)
        np.testing.assert_allclose(gram[1], (C - D) @ (C - D).T, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out[11][2][:4 + l * per], out[11][0][:4 + l * per], rtol=2e-13)
    ok = ~np.isnan(out[11][1])
    np.testing.assert_allclose(out[11][3][ok], out[11][1][ok], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("l", [1, 3])
def test_values_cross_the_boundary_into_every_layout(contexts, l):
    """vs_upload_values / vs_download_values convert on the device: what lands is the host's statement of the layout"""
    from varsens_amd import _native
    k, n = 5, 4700
    rng = np.random.RandomState(l)
    vals = rng.standard_normal((2 * n * (k + 1), l))
    vals[4100, 0] = np.nan
    for sh, ctx in contexts.items():
        f = ctx.alloc(8 * vals.size)
        assert ctx.upload_values(vals, n, k, 0, n, f) == 1
        raw = ctx.download(f, (vals.size,))
        assert np.array_equal(_bits(raw), _bits(_native.to_device_layout(vals.reshape(2 * k + 2, n, l), sh)))
        assert np.array_equal(_bits(ctx.download_values(f, n, k, l)), _bits(vals))
        # a shard: rows [600, 600 + 4099) of every block
        f2 = ctx.alloc(8 * 2 * (k + 1) * 4099 * l)
        assert ctx.upload_values(vals, n, k, 600, 4099, f2, 2.0) == (1 if 600 <= 4100 - 0 * n < 600 + 4099 else 0)
        want = vals.reshape(2 * k + 2, n, l)[:, 600:600 + 4099] / 2.0
        assert np.array_equal(_bits(ctx.download(f2, (want.size,))), _bits(_native.to_device_layout(want, sh)))
        f.free()
        f2.free()


def test_whole_analysis_is_the_same_in_every_layout(contexts):
    """Sample -> Objective -> Varsens with the values in HBM (fused=False), a host callable, and loaded values with NaN
    rows: identical result bits whatever the row-block size"""
    import varsens_amd as vs
    k, n = 6, 4500
    a = [0, 0.5, 3, 9, 99, 99]
    res = {}
    for sh, ctx in contexts.items():
        v = vs.Varsens(vs.objectives.GFunction(a), vs.scale.Identity(), k, n, verbose=False, fused=False, context=ctx)
        s = vs.Sample(k, n, vs.scale.Identity(), verbose=False, context=ctx)
        h = vs.Varsens(vs.Objective(k, n, s, lambda X: X[:, 0] + X[:, 1] * X[:, 2], verbose=False, vectorized=True),
                       verbose=False)
        vals = v.objective.flat().copy()
        vals[4200, 0] = np.nan
        w = vs.Varsens(vs.Objective(k, n, objective_vals=vals, verbose=False, context=ctx), verbose=False)
        res[sh] = [getattr(x, name) for x in (v, h, w) for name in ("E_2", "var_y", "sens", "sens_t", "sens_2", "sens_2n")]
        assert w.objective.fM_1.shape == (n - 1, 1)
    for sh in SHIFTS[1:]:
        for p, q in zip(res[SHIFTS[0]], res[sh]):
            assert np.array_equal(_bits(p), _bits(q)), sh
