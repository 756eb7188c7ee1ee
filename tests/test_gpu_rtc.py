"""User objectives compiled with hiprtc into the fused generate->evaluate kernel."""
import numpy as np
import pytest

from oracle import saltelli_oracle as orc

pytestmark = pytest.mark.gpu

A6 = [0, 0.5, 3, 9, 99, 99]
G_BODY = """
    double acc = 1.0;
    for (int i = 0; i < k; ++i) acc = acc * ((fabs(4.0 * x[i] - 2.0) + p[i]) / (1.0 + p[i]));
    out[0] = acc;
"""


@pytest.fixture(scope="module")
def vs():
    import varsens_amd
    return varsens_amd


def test_expression_equals_builtin_gfunction_bit_for_bit(vs):
    k, n = 6, 4096
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    s._ctx.set_eval_order("strict")     # the built-in then multiplies left to right, like the user's loop
    try:
        o1 = vs.Objective(k, n, s, vs.objectives.GFunction(A6), verbose=False)
    finally:
        s._ctx.set_eval_order("fast")
    o2 = vs.Objective(k, n, s, vs.objectives.Expression(G_BODY, params=A6), verbose=False)
    assert np.array_equal(o1.flat(), o2.flat())
    S = orc.build_sample(k, n)
    F = orc.evaluate_rows(lambda X: orc.g_function_rows(X, A6), *S)
    assert np.array_equal(o2.flat(), orc.flat_objective(*F))
    o3 = vs.Objective(k, n, s, vs.objectives.GFunction(A6), verbose=False)      # default order: scan kernels
    assert np.array_equal(o3.flat(), orc.flat_objective(*orc.evaluate_engine_order("g_function", A6, S[0], S[1])))
    assert np.max(np.abs(o3.flat() - o2.flat()) / o2.flat()) <= 1e-15


def test_expression_ishigami_matches_builtin_and_truth(vs):
    body = """
        const double s1 = sin(x[0]), s2 = sin(x[1]), x3 = x[2];
        out[0] = (s1 + p[0] * (s2 * s2)) + (p[1] * ((x3 * x3) * (x3 * x3))) * s1;
    """
    sc = vs.scale.Linear([-np.pi] * 3, [np.pi] * 3)
    # (fused=False: the built-in objective's default call is its fused kernel, whose sums come in another order)
    v1 = vs.Varsens(vs.objectives.Ishigami(7.0, 0.1), sc, 3, 1 << 16, verbose=False, fused=False)
    v2 = vs.Varsens(vs.objectives.Expression(body, params=[7.0, 0.1]), sc, 3, 1 << 16, verbose=False)
    assert np.array_equal(v1.objective.flat(), v2.objective.flat())
    assert np.array_equal(v1.sens, v2.sens) and np.array_equal(v1.sens_2, v2.sens_2)
    v0 = vs.Varsens(vs.objectives.Ishigami(7.0, 0.1), sc, 3, 1 << 16, verbose=False)
    assert v0.objective._f is None
    for f in ("sens", "sens_t", "sens_2", "sens_2n"):
        np.testing.assert_allclose(getattr(v0, f), getattr(v2, f), rtol=1e-9, atol=1e-11, err_msg=f)
    V, S, ST = orc.ishigami_truth()
    np.testing.assert_allclose(v2.sens[:, 0], S, atol=0.01)
    np.testing.assert_allclose(v2.sens_t[:, 0], ST, atol=0.01)


def test_expression_two_outputs_and_large_k(vs):
    k, n = 70, 300      # 70 KiB of LDS per workgroup: the > 64 KiB dynamic-LDS path
    body = """
        double s = 0.0, t = 0.0;
        for (int i = 0; i < k; ++i) { s = s + p[i] * x[i]; t = t + x[i] * x[i]; }
        out[0] = s; out[1] = t;
    """
    c = np.arange(1, k + 1, dtype=float)
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    o = vs.Objective(k, n, s, vs.objectives.Expression(body, outputs=2, params=c), verbose=False)
    S = orc.build_sample(k, n)
    rows = lambda X: np.stack([orc.linear_rows(X, c), orc.linear_rows(X * X, np.ones(k))], axis=-1)
    F = orc.evaluate_rows(rows, *S)
    assert np.array_equal(o.flat(), orc.flat_objective(*F))
    v = vs.Varsens(o, verbose=False)
    assert v.sens.shape == (k, 2) and v.sens_2.shape == (k, 2, k, 2)


def test_expression_beyond_lds_budget(vs):
    """k = 200: the coordinate tile of a 64-row block is 200 KiB, so the compiled kernel keeps it in HBM scratch."""
    k, n = 200, 150
    a = np.linspace(0.0, 50.0, k)
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    s._ctx.set_eval_order("strict")     # the built-in multiplies left to right, like the user's loop
    try:
        o1 = vs.Objective(k, n, s, vs.objectives.GFunction(a), verbose=False)
    finally:
        s._ctx.set_eval_order("fast")
    o2 = vs.Objective(k, n, s, vs.objectives.Expression(G_BODY, params=a), verbose=False)
    assert np.array_equal(o1.flat(), o2.flat())


@pytest.mark.parametrize("k,n", [(6, 4096), (32, 1000), (45, 300), (3, 5)])
def test_separable_expression_equals_builtins_bit_for_bit(vs, k, n):
    """A user term function compiled into the chain-scheme kernel: same bits as the built-in objectives."""
    rng = np.random.RandomState(k)
    a = rng.rand(k) * 9.0
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    built = vs.Objective(k, n, s, vs.objectives.GFunction(a), verbose=False)
    user = vs.Objective(k, n, s, vs.objectives.SeparableExpression(
        "return (fabs(4.0 * x - 2.0) + p[c]) / (1.0 + p[c]);", fold="prod", params=a), verbose=False)
    assert np.array_equal(built.flat(), user.flat())
    built = vs.Objective(k, n, s, vs.objectives.Linear(a), verbose=False)
    user = vs.Objective(k, n, s, vs.objectives.SeparableExpression("return p[c] * x;", fold="sum", params=a),
                        verbose=False)
    assert np.array_equal(built.flat(), user.flat())
    v = vs.Varsens(user, verbose=False)
    assert v.sens.shape == (k, 1)


@pytest.mark.parametrize("k,n,fold", [(6, 51200, "prod"), (32, 20001, "sum"), (17, 64, "prod"), (16, 3000, "sum"),
                                      (40, 3000, "sum"), (64, 2000, "prod"), (100, 1500, "sum"), (128, 700, "prod")])
def test_separable_expression_default_call_is_one_kernel(vs, k, n, fold):
    """A user term function with k <= 128 through Varsens(...) with its second-order indices: the run-time compiled
    fused evaluate + both-orders kernel (vs_user_gram), nothing written to HBM; results equal the built-in
    objective's fused kernel (same term, same kernel body) and the two-kernel route."""
    rng = np.random.RandomState(k + n)
    a = rng.rand(k) * 5.0
    if fold == "prod":
        user = vs.objectives.SeparableExpression("return (fabs(4.0 * x - 2.0) + p[c]) / (1.0 + p[c]);", fold="prod", params=a)
        built = vs.objectives.GFunction(a)
    else:
        user = vs.objectives.SeparableExpression("return p[c] * x;", fold="sum", params=a)
        built = vs.objectives.Linear(a)
    sc = vs.scale.Linear(-1.0, 2.0) if fold == "sum" else vs.scale.Identity()
    u = vs.Varsens(user, sc, k, n, verbose=False, fused=True)
    assert u.objective._f is None
    b = vs.Varsens(built, sc, k, n, verbose=False, fused=True)
    t = vs.Varsens(user, sc, k, n, verbose=False, fused=False)
    assert t.objective._f is not None
    for f in ("E_2", "var_y", "U_j", "U_nj"):
        np.testing.assert_allclose(getattr(u, f), getattr(t, f), rtol=1e-12, err_msg=f)
    for f in ("sens", "sens_t", "sens_2", "sens_2n"):
        np.testing.assert_allclose(getattr(u, f), getattr(t, f), rtol=1e-9, atol=1e-11, err_msg=f)
        # the g-function's built-in term takes its quotient from a rounded reciprocal (same bits as the division)
        np.testing.assert_allclose(getattr(u, f), getattr(b, f), rtol=1e-9, atol=1e-11, err_msg=f)
    # power scaling has no run-time compiled fused kernel: the two-kernel route by itself
    w = vs.Varsens(user, vs.scale.Power(1.0, 10.0), k, min(n, 2000), verbose=False)
    assert w.objective._f is not None


@pytest.mark.parametrize("k,n,fold", [(6, 5000, "prod"), (32, 3001, "sum"), (48, 2000, "prod"), (64, 1500, "sum"),
                                      (100, 1200, "prod"), (128, 700, "sum"), (200, 300, "sum")])
def test_separable_expression_on_a_sobol_design(vs, k, n, fold):
    """The run-time compiled scan kernels take Sobol points from nibble tables passed through LDS (round 5: a KiB per
    wave added to the launch for a Sobol design): evaluation bit for bit the built-in objective's on the same design,
    the fused first-order and both-orders kernels against the two-kernel route."""
    rng = np.random.RandomState(k + n)
    a = rng.rand(k) * 5.0
    if fold == "prod":
        user = vs.objectives.SeparableExpression("return (fabs(4.0 * x - 2.0) + p[c]) / (1.0 + p[c]);", fold="prod", params=a)
        built = vs.objectives.GFunction(a)
    else:
        user = vs.objectives.SeparableExpression("return p[c] * x;", fold="sum", params=a)
        built = vs.objectives.Linear(a)
    sc = vs.scale.Linear(0.1, 0.9)
    s = vs.Sample(k, n, sc, verbose=False, generator='sobol')
    ou = vs.Objective(k, n, s, user, verbose=False)
    ob = vs.Objective(k, n, s, built, verbose=False)
    assert np.array_equal(ou.flat(), ob.flat())
    t = vs.Varsens(ou, verbose=False)                         # evaluation + reduction
    if k <= 128:
        u = vs.Varsens(user, sc, k, n, sample=vs.Sample(k, n, sc, verbose=False, generator='sobol'), verbose=False, fused=True)
        assert u.objective._f is None                         # one kernel, no f-values
        for f in ("E_2", "var_y", "U_j", "U_nj"):
            np.testing.assert_allclose(getattr(u, f), getattr(t, f), rtol=1e-11, err_msg=f)
        for f in ("sens", "sens_t", "sens_2", "sens_2n"):
            np.testing.assert_allclose(getattr(u, f), getattr(t, f), rtol=1e-8, atol=1e-10, err_msg=f)
        w = vs.Varsens(user, sc, k, n, sample=vs.Sample(k, n, sc, verbose=False, generator='sobol'), verbose=False, fused=True,
                       second_order=False)
        np.testing.assert_allclose(w.sens, t.sens, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(w.sens_t, t.sens_t, rtol=1e-8, atol=1e-10)


def test_expression_compile_error_is_reported(vs):
    from varsens_amd._native import VarsensHipError
    s = vs.Sample(3, 8, vs.scale.Identity(), verbose=False)
    with pytest.raises(VarsensHipError, match="does not compile"):
        vs.Objective(3, 8, s, vs.objectives.Expression("out[0] = not_a_symbol(x[0]);"), verbose=False)
