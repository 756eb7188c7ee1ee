"""Objective values and sample rows across the host boundary (csrc/vs_values.hip): the routes of a reference user whose
model runs elsewhere -- Objective(k, n, objective_vals=v) (varsens/saltelli.py:297-298, 415-495), Objective.flat()
(:357-391) and the row-by-row evaluation loops (:308-355).  The device lays the values out and counts the NaN rows; the
expected side is numpy on the host (the layout) and the oracle (the numbers)."""
import numpy
import pytest

from oracle import saltelli_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vs():
    import varsens_amd
    return varsens_amd


@pytest.fixture(scope="module")
def ctx(vs):
    from varsens_amd.saltelli import default_context
    return default_context()


def _bits(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float64).view(numpy.uint64)


def _values(k, n, l, seed, nans=()):
    rng = numpy.random.default_rng(seed)
    v = rng.standard_normal((2 * n * (k + 1), l)) * 10.0 ** rng.integers(-3, 4, size=(2 * n * (k + 1), l))
    for (row, out) in nans:
        v[row, out] = numpy.nan
    return v


def _device(ctx, vals, k, n, l, begin, rows):
    """what the device buffer must hold: the shard's values in the row-blocked layout (include/varsens_hip.h), laid out
    by numpy on the host"""
    from varsens_amd import _native
    blocks = vals.reshape(2 * k + 2, n, l)[:, begin:begin + rows]
    return _native.to_device_layout(blocks, ctx.f_shift)


# (k, n, l, shard): one output (the copy engine writes the destination), a few, more than a tile of 32, row counts off
# the 128-row tile, a true shard, pieces beyond one ring slot (16 MiB) and beyond all three
SHAPES = [(6, 1024, 1, None), (3, 7, 1, None), (5, 64, 2, None), (7, 50, 3, None), (4, 129, 5, None), (3, 300, 33, None),
          (2, 77, 70, None), (6, 1000, 3, (123, 456)), (6, 1000, 1, (0, 1)), (5, 40, 4, (39, 1)),
          (32, 1 << 15, 1, None), (6, 1 << 19, 3, None), (6, 1 << 17, 3, (5, (1 << 17) - 9))]


@pytest.mark.parametrize("k,n,l,shard", SHAPES)
def test_upload_values_lays_out_what_numpy_would(vs, ctx, k, n, l, shard):
    begin, rows = shard or (0, n)
    vals = _values(k, n, l, seed=k * 1000 + l)
    f = ctx.alloc(8 * 2 * (k + 1) * rows * l)
    lost = ctx.upload_values(vals, n, k, begin, rows, f)
    assert lost == 0
    got = ctx.download(f, (l * (2 * k + 2) * rows,))
    assert numpy.array_equal(_bits(got), _bits(_device(ctx, vals, k, n, l, begin, rows)))
    # and back: Objective.flat() of the shard, the reference's order
    back = ctx.download_values(f, rows, k, l)
    want = vals.reshape(2 * k + 2, n, l)[:, begin:begin + rows].reshape(-1, l)
    assert numpy.array_equal(_bits(back), _bits(want))
    f.free()


@pytest.mark.parametrize("l,divide", [(1, 4.0), (1, numpy.array([3.0])), (3, 7.0), (3, numpy.array([3.0, -0.1, 1e300])),
                                      (40, numpy.arange(1.0, 41.0))])
def test_upload_values_divides_like_numpy(vs, ctx, l, divide):
    """`x / scaling` of Objective.load (varsens/saltelli.py:451-468): the device's quotient has numpy's bits"""
    k, n = 4, 333
    vals = _values(k, n, l, seed=l)
    f = ctx.alloc(8 * 2 * (k + 1) * n * l)
    ctx.upload_values(vals, n, k, 0, n, f, divide)
    got = ctx.download(f, (l * (2 * k + 2) * n,))
    assert numpy.array_equal(_bits(got), _bits(_device(ctx, vals / divide, k, n, l, 0, n)))
    f.free()


def test_nan_rows_are_counted_on_the_device(vs, ctx):
    """Objective.load deletes row r when ANY block holds a NaN at r in output 0 (varsens/saltelli.py:474-486): the count
    comes back from vs_upload_values; NaNs in other outputs do not count, two NaNs in one row count once, and a shard
    counts its own rows only."""
    k, n, l = 5, 200, 3
    N = 2 * n * (k + 1)
    nans = [(3, 0), (n + 3, 0),          # row 3 twice (fM_1 and fM_2)
            (2 * n + 17, 0),             # row 17 in fN_j[0]
            ((2 + k) * n + 150, 0),      # row 150 in fN_nj[0]
            (5 * n + 199, 0),            # row 199
            (40, 2), (n + 41, 1)]        # other outputs: no row lost
    vals = _values(k, n, l, seed=5, nans=nans)
    assert int(orc.nan_row_mask(vals, k, n).sum()) == 4
    f = ctx.alloc(8 * N * l)
    assert ctx.upload_values(vals, n, k, 0, n, f) == 4
    assert ctx.upload_values(vals, n, k, 0, 100, f) == 2          # rows 3 and 17
    assert ctx.upload_values(vals, n, k, 100, 100, f) == 2        # rows 150 and 199
    assert ctx.upload_values(vals, n, k, 18, 100, f) == 0
    f.free()


@pytest.mark.parametrize("n", [64, 1 << 14])
def test_three_outputs_with_nans_against_the_oracle(vs, n, capsys):
    """Call stack (c) end to end with l = 3 and NaNs in outputs 0 and 2 (varsens/saltelli.py:449-495, 572-622):
    Objective(k, n, objective_vals=v) -> Varsens(o) against the oracle's load_objective + compute_varsens.  Output 0's
    NaN rows are deleted from every output; output 2's own NaNs stay and poison output 2 alone."""
    k, l = 6, 3
    X = orc.build_sample(k, n)
    a = [[0, 0.5, 3, 9, 99, 99], [99, 99, 9, 3, 0.5, 0], [1, 1, 1, 1, 1, 1]]
    F = [numpy.stack([orc.g_function_rows(x.reshape(-1, k), numpy.array(ai)).reshape(x.shape[:-1]) for ai in a], axis=-1)
         for x in X]
    vals = orc.flat_objective(*F).copy()
    N = vals.shape[0]
    vals[5, 0] = numpy.nan
    vals[n + 9, 0] = numpy.nan
    vals[3 * n + 5, 0] = numpy.nan         # row 5 again, another block
    vals[(2 + k) * n + 33, 0] = numpy.nan
    o = vs.Objective(k, n, objective_vals=vals, verbose=False)
    assert "3 of %d objectives were NaN" % N in capsys.readouterr().out
    want = orc.load_objective(vals, k, n)
    assert want[4] == 3
    assert o.fM_1.shape == (n - 3, l)
    for got, ref in zip((o.fM_1, o.fM_2, o.fN_j, o.fN_nj), want[:4]):
        assert numpy.array_equal(_bits(got), _bits(ref))
    v = vs.Varsens(o, verbose=False)
    R = orc.compute_varsens(*want[:4], n)
    for name in ("E_2", "var_y", "U_j", "U_nj"):
        numpy.testing.assert_allclose(getattr(v, name), R[name], rtol=1e-12)
    for name in ("sens", "sens_t", "sens_2", "sens_2n"):
        numpy.testing.assert_allclose(getattr(v, name), R[name], rtol=1e-9, atol=1e-11)
    # now a NaN in output 2 that output 0 does not share: the row stays, output 2's numbers are NaN, the others' are not
    vals[7, 2] = numpy.nan
    o2 = vs.Objective(k, n, objective_vals=vals, verbose=False)
    v2 = vs.Varsens(o2, verbose=False)
    assert o2.fM_1.shape == (n - 3, l)
    numpy.testing.assert_allclose(v2.sens[:, :2], R["sens"][:, :2], rtol=1e-9, atol=1e-11)
    assert numpy.isnan(v2.sens[:, 2]).all() and numpy.isnan(v2.E_2[2])
    numpy.testing.assert_allclose(v2.sens_2[:, :2, :, :2], R["sens_2"][:, :2, :, :2], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("k,l,n", [(40, 1, 2064), (56, 1, 4110), (64, 1, 2048 * 2 + 40), (33, 1, 50), (20, 3, 1000), (16, 4, 2050),
                                   (48, 1, 30000)])
def test_mid_sized_designs_against_the_oracle(vs, k, l, n):
    """Call stack (c) at 33 .. 64 index rows (csrc/vs_gram_dma.hip: one output -- both orders in one kernel, the first-order
    dot products by row-owner lanes; several outputs -- second order alone beside the first-order kernel): values from a
    model that ran elsewhere -> Objective(k, n, objective_vals=v) -> Varsens(o), every result of varsens/saltelli.py:572-622
    against the oracle's load_objective + compute_varsens on the same values."""
    rng = numpy.random.default_rng(k * 1000 + l * 10 + n)
    N = 2 * n * (k + 1)
    # a linear model with interactions: values with structure in every index row, not noise around zero
    base = rng.standard_normal((2 * k + 2, n, 1))
    vals = (3.0 + base + 0.25 * rng.standard_normal((2 * k + 2, n, l)) * numpy.arange(1, l + 1)).reshape(N, l)
    o = vs.Objective(k, n, objective_vals=vals, verbose=False)
    v = vs.Varsens(o, verbose=False)
    want = orc.load_objective(vals, k, n)
    R = orc.compute_varsens(*want[:4], n)
    for name in ("E_2", "var_y", "U_j", "U_nj"):
        numpy.testing.assert_allclose(getattr(v, name), R[name], rtol=1e-12)
    scale = numpy.abs(R["sens_2"]).max()
    for name in ("sens", "sens_t"):
        numpy.testing.assert_allclose(getattr(v, name), R[name], rtol=1e-10, atol=1e-12)
    for name in ("sens_2", "sens_2n"):
        numpy.testing.assert_allclose(getattr(v, name), R[name], rtol=1e-9, atol=1e-11 * max(1.0, scale))


def test_load_divides_per_output_and_falls_back_for_other_shapes(vs):
    """Objective.load(..., scaling=s): a scalar and one divisor per output are divided on the device, anything else
    numpy can broadcast against (N, l) on the host -- the values are the reference's x / scaling in every case"""
    k, n, l = 4, 40, 3
    vals = _values(k, n, l, seed=11)
    for scaling in (2.5, numpy.array([2.0, 3.0, 5.0]), numpy.array([[2.0, 3.0, 5.0]]),
                    numpy.arange(1.0, 1.0 + vals.shape[0])[:, None]):
        o = vs.Objective(k, n, verbose=False, obj_vals=vals, scaling=scaling)
        assert numpy.array_equal(_bits(o.flat()), _bits(vals / scaling))


def test_sample_rows_equal_sample_flat(vs, ctx):
    """vs_sample_rows (pieces through the pinned ring) against vs_sample_flat + one download, and the oracle"""
    k, n = 7, 3000
    s = vs.Sample(k, n, vs.scale.Linear([-1.0] * k, [2.0] * k), verbose=False)
    total = 2 * n * (k + 1)
    want = orc.flat_sample(*orc.build_sample(k, n, lambda x: orc.scale_linear(x, -1.0, 2.0)))
    for begin, count in [(0, total), (0, 1), (total - 1, 1), (n - 3, 7), (2 * n + 5, 3 * n), (17, 0)]:
        got = ctx.sample_rows(s._design, begin, count, k)
        assert got.shape == (count, k)
        assert numpy.array_equal(got, want[begin:begin + count])
    # more than three ring slots of 16 MiB: k = 64 columns x 2^17 rows = 64 MiB
    big = vs.Sample(64, 1 << 10, vs.scale.Identity(), verbose=False)
    rows = ctx.sample_rows(big._design, 0, 1 << 17, 64)
    buf = ctx.alloc(8 * 64 << 17)
    ctx.sample_flat(big._design, 0, 1 << 17, buf)
    assert numpy.array_equal(rows, ctx.download(buf, (1 << 17, 64)))
    buf.free()


def test_sample_row_chunks_overlap_and_stay_valid_while_used(vs, ctx):
    """vs_sample_rows_begin / _wait: chunk i + 1 is produced while chunk i is in the caller's hands; a view is valid
    until the next one is asked for"""
    k, n = 5, 1000
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    want = orc.flat_sample(*orc.build_sample(k, n))
    spans = [(0, 10), (10, 990), (1000, 1), (1001, 4000), (5001, 3), (2 * n * (k + 1) - 50, 50)]
    seen = 0
    for (begin, count), view in zip(spans, s._row_chunks(spans)):
        assert view.shape == (count, k) and not view.flags.writeable
        assert numpy.array_equal(view, want[begin:begin + count])
        seen += 1
    assert seen == len(spans)
    with pytest.raises(Exception, match="nothing was begun"):
        ctx.sample_rows_wait(0, 1, k)
    # a slot begun twice without a wait in between: the first copy is let through, the second one is what comes back
    ctx.sample_rows_begin(s._design, 0, 5, 1)
    ctx.sample_rows_begin(s._design, 7, 5, 1)
    assert numpy.array_equal(ctx.sample_rows_wait(1, 5, k), want[7:12])


def test_row_by_row_model_may_keep_its_rows(vs):
    """The reference hands sample.M_1[i] to the objective (varsens/saltelli.py:335-355); a model that keeps the rows it
    was given (logging, caching) must still hold them afterwards -- the row-by-row path copies each chunk out of the
    library's pinned slot"""
    import varsens_amd.saltelli as S
    k, n = 3, 40
    kept = []

    def model(x):
        kept.append(x)
        return float(x.sum())
    old = S._HOST_CHUNK_ROWS
    S._HOST_CHUNK_ROWS = 16      # several chunks per block, both slots reused many times
    try:
        v = vs.Varsens(model, vs.scale.Identity(), k, n, verbose=False)
    finally:
        S._HOST_CHUNK_ROWS = old
    flat = orc.flat_sample(*orc.build_sample(k, n))
    assert len(kept) == 1 + 2 * n * (k + 1)
    assert numpy.array_equal(numpy.stack(kept[1:]), flat)
    assert numpy.array_equal(v.objective.flat()[:, 0], flat.sum(axis=1))


def test_context_close_releases_its_designs(vs):
    """A design belongs to the context that made it; vs_context_destroy refuses while one is alive, Context.close()
    releases them first (include/varsens_hip.h, ownership)."""
    import ctypes
    from varsens_amd import _native
    c = _native.Context(0)
    s = vs.Sample(4, 32, vs.scale.Identity(), verbose=False, context=c)
    assert c.lib.vs_context_destroy(c.handle) == -5                      # VS_ERR_STATE
    assert b"still alive" in c.lib.vs_last_error()
    assert c.lib.vs_design_destroy(None, s._design.h) == 0               # released through its owner, whatever ctx says
    s._design.h = None
    t = vs.Sample(4, 32, vs.scale.Identity(), verbose=False, context=c)
    assert t._design.h
    c.close()
    assert c.handle is None and t._design.h is None


def test_sample_can_leave_numpy_seeded_as_the_reference_does(vs):
    """varsens/saltelli.py:100: numpy.random.seed(1) + the shuffle of M_2 are a visible side effect on numpy's global
    generator; opt-in here (reseed_numpy=True or VARSENS_RESEED_NUMPY=1), the default leaves the generator alone."""
    k, n = 4, 500
    saved = numpy.random.get_state()
    try:
        numpy.random.seed(1)
        numpy.random.shuffle(numpy.zeros((n, k)))
        want = numpy.random.random(3)
        numpy.random.seed(777)
        before = numpy.random.get_state()[1].copy()
        vs.Sample(k, n, vs.scale.Identity(), verbose=False)
        assert numpy.array_equal(numpy.random.get_state()[1], before)
        vs.Sample(k, n, vs.scale.Identity(), verbose=False, reseed_numpy=True)
        assert numpy.array_equal(numpy.random.random(3), want)
    finally:
        numpy.random.set_state(saved)


def test_values_in_and_out_over_random_shapes(vs, ctx):
    """vs_upload_values / vs_download_values at random (k, n, l, shard, divisor): rows around the row-block size of the
    layout and around the 128-row tile of the layout kernel, pieces around the 16 MiB ring slot -- always the layout numpy
    builds, always the reference's order back"""
    rng = numpy.random.default_rng(2024)
    step = 1 << ctx.f_shift
    for trial in range(40):
        k = int(rng.integers(1, 12))
        l = int(rng.choice([1, 1, 2, 3, 7, 33]))
        n = int(rng.choice([1, 2, 127, 128, 129, step - 1, step, step + 1, 2 * step + 5, int(rng.integers(1, 3 * step))]))
        if rng.random() < 0.5:
            begin = int(rng.integers(0, n))
            rows = int(rng.integers(0, n - begin + 1))
        else:
            begin, rows = 0, n
        divide = None if rng.random() < 0.5 else (float(rng.uniform(0.5, 4.0)) if rng.random() < 0.5 or l == 1
                                                  else rng.uniform(0.5, 4.0, l))
        vals = rng.standard_normal((2 * n * (k + 1), l))
        nan_rows = set()
        for _ in range(int(rng.integers(0, 4))):
            r = int(rng.integers(0, n))
            vals[int(rng.integers(0, 2 * k + 2)) * n + r, 0] = numpy.nan
            nan_rows.add(r)
        f = ctx.alloc(8 * 2 * (k + 1) * max(rows, 1) * l)
        lost = ctx.upload_values(vals, n, k, begin, rows, f, divide)
        assert lost == sum(1 for r in nan_rows if begin <= r < begin + rows), (trial, k, n, l, begin, rows)
        if rows:
            want = vals if divide is None else vals / divide
            got = ctx.download(f, (l * (2 * k + 2) * rows,))
            assert numpy.array_equal(_bits(got), _bits(_device(ctx, want, k, n, l, begin, rows))), (trial, k, n, l, begin, rows)
            back = ctx.download_values(f, rows, k, l)
            assert numpy.array_equal(_bits(back), _bits(want.reshape(2 * k + 2, n, l)[:, begin:begin + rows].reshape(-1, l)))
        f.free()
