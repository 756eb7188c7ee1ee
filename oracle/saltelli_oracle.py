"""CPU oracle for the Saltelli hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  ``varsens_amd`` never does: the product
path runs on the HIP library or fails.

It is a numpy (Python 3) restatement of what LoLab-MSM/varsens computes on
this path.  Every function cites the reference file:line it follows
(paths relative to the reference checkout).

Pinning status
--------------
* Everything downstream of the unit-cube points (scaling, seed-1 shuffle,
  N_j/N_nj construction, flat order, objective slicing + NaN trim, E_2, var_y,
  U_j, U_nj, sens, sens_t, sens_2, sens_2n) is PINNED: ``tests/golden/*.npz``
  hold outputs of the reference's own ``saltelli.py``/``scale.py`` executed in
  the build container (generator: ``tests/golden/make_golden.py``) and
  ``tests/test_oracle_golden.py`` checks this module against them.
* Halton point VALUES: **parity unpinned**.  The reference draws them from the
  third-party ``ghalton`` C++/SWIG module (``varsens/saltelli.py:1,82-84``),
  which is not in the reference tree, is version-unpinned
  (``doc/requirements.txt:3``) and is not installed here.  The reference's own
  tests only pin range and mean (``varsens/tests/test_sample.py:23-64``).
  The definition used here (and by the HIP kernel, bit for bit): bases are the
  first k primes, the first point of the stream has index 1, point i in base
  b is the digit-reversed integer of i divided by b**digits -- one correctly
  rounded fp64 division of two exact integers.  It agrees with
  ``scipy.stats.qmc.Halton(scramble=False)`` (which accumulates digit * b**-j in floating point) to
  <= 2 ulp over the first thousands of points in up to 128 dimensions, <= 3 ulp seen at indices near
  1e6, about half of the values bit-identical (tests/test_oracle_golden.py::
  test_halton_agrees_with_scipy_to_two_ulp; the device against scipy directly in
  tests/test_gpu_kernels.py::test_device_halton_against_scipy).  That is the size of the
  difference to expect from ANY other plain Halton implementation, ghalton included.
* Sobol point values: **parity unpinned** against ``quantlib/sobolGen.cpp``
  (needs QuantLib + its Levitan-Lemieux direction integers, neither present).
  Pinned instead, bit for bit, to ``scipy.stats.qmc.Sobol(scramble=False)``
  (Joe-Kuo direction numbers).
"""
import math

import numpy as np

# ----------------------------------------------------------------------------
# a1: low-discrepancy generators
# ----------------------------------------------------------------------------


def first_primes(k):
    """First k primes (Halton bases; ghalton.Halton(k), varsens/saltelli.py:82)."""
    out = []
    c = 2
    while len(out) < k:
        if all(c % p for p in out if p * p <= c):
            out.append(c)
        c += 1
    return np.array(out, dtype=np.int64)


def halton_skip(k, discard=0):
    """Index of the first point the reference keeps.

    ``seq.get(20*k + discard)`` throws away the leading points
    (varsens/saltelli.py:83); the stream's first point has index 1.
    """
    return 1 + 20 * int(k) + int(discard)


def halton(k, start, count):
    """Halton points for indices start .. start+count-1, shape (count, k).

    Stands for ``ghalton.Halton(k).get(...)`` (varsens/saltelli.py:82-84).
    value = (sum_j d_j b^(m-1-j)) / b^m with i = sum_j d_j b^j, m = #digits.
    Both integers are exact in fp64 while start+count < 2^53 / b.
    """
    idx = np.arange(start, start + count, dtype=np.int64)
    out = np.empty((count, k), dtype=np.float64)
    for c, b in enumerate(first_primes(k)):
        b = int(b)
        q = idx.copy()
        num = np.zeros_like(idx)
        den = np.ones_like(idx)
        while True:
            live = q > 0
            if not live.any():
                break
            d = q % b
            num = np.where(live, num * b + d, num)
            den = np.where(live, den * b, den)
            q = q // b
        out[:, c] = num.astype(np.float64) / den.astype(np.float64)
    return out


def sobol(k, start, count, direction):
    """Sobol (Joe-Kuo) points for indices start.. (index 0 is the origin).

    ``direction`` is the (k, 32) uint32 table of direction integers v[c][j]
    (already shifted left, bit 31 = 2^-1).  Point i = XOR over set bits j of the
    Gray code g = i ^ (i >> 1) of v[c][j]; value = x * 2^-32 (exact in fp64).
    Capability counterpart of quantlib/sobolGen.cpp:47-60 (values unpinned to
    it, pinned to scipy instead -- see module docstring).
    """
    idx = np.arange(start, start + count, dtype=np.uint64)
    gray = idx ^ (idx >> np.uint64(1))
    out = np.empty((count, k), dtype=np.float64)
    for c in range(k):
        x = np.zeros(count, dtype=np.uint64)
        for j in range(32):
            bit = (gray >> np.uint64(j)) & np.uint64(1)
            x ^= bit * np.uint64(int(direction[c, j]))
        out[:, c] = x.astype(np.float64) * (1.0 / 4294967296.0)
    return out


# ----------------------------------------------------------------------------
# a2: scaling (varsens/scale.py)
# ----------------------------------------------------------------------------


def scale_linear(points, lower, upper):
    """varsens/scale.py:33 -- multiply then add, two roundings (no fma)."""
    return points * (upper - lower) + lower


def scale_power(points, lower, upper):
    """varsens/scale.py:62."""
    return lower * ((upper / lower) ** points)


def scale_percentage(points, reference, percentage=50.0):
    """varsens/scale.py:90-91."""
    diff = percentage * reference / 100.0
    return scale_linear(points, reference - diff, reference + diff)


def scale_magnitude(points, reference, orders=3.0, base=10.0):
    """varsens/scale.py:121-122."""
    factor = base ** orders
    return scale_power(points, reference / factor, reference * factor)


# ----------------------------------------------------------------------------
# a3: the seed-1 de-correlation shuffle
# ----------------------------------------------------------------------------


def seed1_permutation(n):
    """perm with shuffled_M_2[r] = M_2[perm[r]] (varsens/saltelli.py:100-101).

    The reference reseeds numpy's global legacy RNG to 1 and row-shuffles the
    (n,k) array in place; shuffling arange(n) with a private RandomState(1)
    consumes the same MT19937 draws and yields the same row order (pinned by
    golden G7).
    """
    p = np.arange(n, dtype=np.int64)
    np.random.RandomState(1).shuffle(p)
    return p


# ----------------------------------------------------------------------------
# a1-a5: Sample
# ----------------------------------------------------------------------------


def column_swap(base_rows, donor_rows):
    """(k,n,k): copy i = base_rows with column i taken from donor_rows.

    varsens/saltelli.py:112-125 (``generate_N_j(M_1, M_2)`` has donor=M_1,
    base=M_2).
    """
    n, k = base_rows.shape
    out = np.broadcast_to(base_rows, (k, n, k)).copy()
    ii = np.arange(k)
    out[ii, :, ii] = donor_rows.T
    return out


def build_sample(k, n, scaling=None, discard=0, raw=None):
    """M_1, M_2, N_j, N_nj exactly as Sample.__init__ (varsens/saltelli.py:62-110).

    ``scaling`` is a callable applied to an (n,k) block, twice (:92,95), or None
    for identity (:86-88).  ``raw`` is an external (2n,k) unit-cube array (:69-73);
    otherwise Halton points from index halton_skip(k, discard) on (:82-84).
    """
    if raw is None:
        x = halton(k, halton_skip(k, discard), 2 * n)
    else:
        x = np.asarray(raw, dtype=np.float64)
        if x.shape != (2 * n, k):
            raise Exception("Raw sample dimensions do not match specified dimensions")
    f = scaling if scaling is not None else (lambda p: p)
    M_1 = np.array(f(x[0:n]), dtype=np.float64)
    M_2 = np.array(f(x[n:2 * n]), dtype=np.float64)[seed1_permutation(n)]
    N_j = column_swap(M_2, M_1)   # :105  N_j[i] = M_2 with column i from M_1
    N_nj = column_swap(M_1, M_2)  # :108  N_nj[i] = M_1 with column i from M_2
    return M_1, M_2, N_j, N_nj


def flat_sample(M_1, M_2, N_j, N_nj):
    """Row order [M_1; M_2; N_j[0..k); N_nj[0..k)] (varsens/saltelli.py:127-160)."""
    k = M_1.shape[1]
    return np.concatenate([M_1, M_2, N_j.reshape(-1, k), N_nj.reshape(-1, k)], axis=0)


# ----------------------------------------------------------------------------
# objectives evaluated on device (defined by this build; the reference takes
# an arbitrary Python callable -- varsens/saltelli.py:308,329-355).  Each is
# written with the exact operation order the HIP kernels use.
# ----------------------------------------------------------------------------


def g_function_rows(X, a):
    """Sobol g-function prod_i (|4x_i-2|+a_i)/(1+a_i), left-to-right product
    (varsens/tests/test_g_function.py:9-14)."""
    X = np.asarray(X, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    fac = (np.abs(4.0 * X - 2.0) + a) / (1.0 + a)
    out = fac[..., 0].copy()
    for i in range(1, X.shape[-1]):
        out = out * fac[..., i]
    return out


def ishigami_rows(X, a=7.0, b=0.1):
    """sin x1 + a sin^2 x2 + b x3^4 sin x1, evaluated as
    (s1 + a*(s2*s2)) + (b*((x3*x3)*(x3*x3)))*s1."""
    X = np.asarray(X, dtype=np.float64)
    s1 = np.sin(X[..., 0])
    s2 = np.sin(X[..., 1])
    x3 = X[..., 2]
    q = (x3 * x3) * (x3 * x3)
    return (s1 + a * (s2 * s2)) + (b * q) * s1


def linear_rows(X, c):
    """sum_i c_i x_i accumulated left to right from 0 with separate mul/add."""
    X = np.asarray(X, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    out = np.zeros(X.shape[:-1], dtype=np.float64)
    for i in range(X.shape[-1]):
        out = out + c[i] * X[..., i]
    return out


def polynomial_rows(X, beta):
    """prod_i (1 + beta_i (x_i - 1/2)), left-to-right product."""
    X = np.asarray(X, dtype=np.float64)
    beta = np.asarray(beta, dtype=np.float64)
    fac = 1.0 + beta * (X - 0.5)
    out = fac[..., 0].copy()
    for i in range(1, X.shape[-1]):
        out = out * fac[..., i]
    return out


# ----------------------------------------------------------------------------
# separable objectives in segmented-scan order (varsens_amd/csrc/vs_device_scan.h)
#
# For k >= 33 columns the HIP engine folds the per-coordinate terms of a separable objective in a
# fixed REASSOCIATED order (O(k) work per design row instead of O(k^2)); the fused evaluate+reduce
# kernels use that order for every k.  The reference has no counterpart: it calls an arbitrary
# Python function row by row (varsens/saltelli.py:329-355), for which any association order of a
# product / sum is as legitimate as numpy.prod's.  The order is restated here exactly, so that the
# device values can be compared bit for bit; against the strict left-to-right folds above they
# differ by rounding only (asserted by the tests at <= 1e-15 relative).
# ----------------------------------------------------------------------------

SCAN_SUB = 4


def scan_shape(k, fused=False):
    """(L, W): columns per segment and segments, as the kernels choose them (vs_scan.hip)."""
    if k <= 8: return 8, 1
    if k <= 16: return 16, 1
    if k <= 24: return 12, 2
    if k <= 32: return 8, 4
    if k <= 48: return 12, 4
    if k <= 64: return 8, 8
    if k <= 96: return 12, 8
    if k <= 128: return 16, 8
    if k <= 256 and not fused: return 32, 8
    raise ValueError("k=%d is outside the scan kernels" % k)


def separable_terms(kind, X, params):
    """Per-coordinate terms t_c(x_c) of the built-in separable objectives, same operations as vs_term."""
    X = np.asarray(X, dtype=np.float64)
    p = np.asarray(params, dtype=np.float64)
    if kind == "g_function":
        return (np.abs(4.0 * X - 2.0) + p) / (1.0 + p)
    if kind == "polynomial":
        return 1.0 + p * (X - 0.5)
    if kind == "linear":
        return p * X
    raise ValueError(kind)


def separable_scan(ta, tb, shape, add=False):
    """a, b (rows,), c, d (k, rows) from term arrays ta, tb (rows, k) in the kernels' order:
    scan position (segment q, place j) holds column j*W + q (VS_SCAN_COLUMN: segments interleave the
    columns); segment totals left to right; prefixes over earlier segments left to right from the identity;
    suffixes over later segments and inside a segment right to left; c_i = (pb . ta_i) . sb_i."""
    L, W = shape
    rows, k = ta.shape
    ident = 0.0 if add else 1.0
    op = (lambda x, y: x + y) if add else (lambda x, y: x * y)
    pad = np.full((rows, L * W - k), ident)
    ta = np.concatenate([ta, pad], axis=1)
    tb = np.concatenate([tb, pad], axis=1)
    column = np.array([j * W + q for q in range(W) for j in range(L)])    # of scan position q*L + j
    ta, tb = ta[:, column], tb[:, column]                                 # everything below is in scan positions
    full = np.full(rows, ident)
    Ta, Tb = [], []
    for q in range(W):
        x, y = full.copy(), full.copy()
        for j in range(L):
            x = op(x, ta[:, q * L + j])
            y = op(y, tb[:, q * L + j])
        Ta.append(x); Tb.append(y)
    a, b = full.copy(), full.copy()
    for q in range(W):
        a = op(a, Ta[q]); b = op(b, Tb[q])
    c = np.empty((k, rows)); d = np.empty((k, rows))
    for q in range(W):
        Pa, Pb = full.copy(), full.copy()
        for w in range(q):
            Pa = op(Pa, Ta[w]); Pb = op(Pb, Tb[w])
        Sa, Sb = full.copy(), full.copy()
        for w in range(W - 1, q, -1):
            Sa = op(Ta[w], Sa); Sb = op(Tb[w], Sb)
        sa = [None] * L; sb = [None] * L          # fold of the terms AFTER column j
        sa[L - 1], sb[L - 1] = Sa, Sb
        for j in range(L - 2, -1, -1):
            sa[j] = op(ta[:, q * L + j + 1], sa[j + 1])
            sb[j] = op(tb[:, q * L + j + 1], sb[j + 1])
        pa, pb = Pa, Pb
        for j in range(L):
            p = q * L + j
            i = column[p]
            if i < k:
                c[i] = op(op(pb, ta[:, p]), sb[j])     # N_j[i]: M_2 with column i from M_1
                d[i] = op(op(pa, tb[:, p]), sa[j])     # N_nj[i]: M_1 with column i from M_2
            pa = op(pa, ta[:, p]); pb = op(pb, tb[:, p])
    return a, b, c, d


def evaluate_scan(kind, params, M_1, M_2, fused=False):
    """fM_1, fM_2 (n,l), fN_j, fN_nj (k,n,l) of a built-in separable objective in the scan kernels'
    order; params (k,) or (l,k)."""
    P = np.atleast_2d(np.asarray(params, dtype=np.float64))
    n, k = M_1.shape
    l = P.shape[0]
    fM_1 = np.empty((n, l)); fM_2 = np.empty((n, l))
    fN_j = np.empty((k, n, l)); fN_nj = np.empty((k, n, l))
    for o in range(l):
        a, b, c, d = separable_scan(separable_terms(kind, M_1, P[o]), separable_terms(kind, M_2, P[o]),
                                    scan_shape(k, fused), add=(kind == "linear"))
        fM_1[:, o], fM_2[:, o], fN_j[:, :, o], fN_nj[:, :, o] = a, b, c, d
    return fM_1, fM_2, fN_j, fN_nj


def evaluate_engine_order(kind, params, M_1, M_2, strict=False):
    """A built-in separable objective in the order the engine's vs_eval uses by default: the scan
    order for k <= 256 (VS_ORDER_FAST), strictly left to right beyond that or when ``strict``
    (VS_ORDER_STRICT, = numpy.prod over the factors as in the reference's tests)."""
    P = np.atleast_2d(np.asarray(params, dtype=np.float64))
    k = M_1.shape[1]
    if k <= 256 and not strict:
        return evaluate_scan(kind, P, M_1, M_2)
    one = {"g_function": g_function_rows, "linear": linear_rows, "polynomial": polynomial_rows}[kind]
    return evaluate_rows(lambda X: np.stack([one(X, p) for p in P], axis=-1), M_1, M_2,
                         column_swap(M_2, M_1), column_swap(M_1, M_2))


# ----------------------------------------------------------------------------
# a6: Objective evaluation, a7: loading pre-computed values
# ----------------------------------------------------------------------------


def evaluate_rows(rows_func, M_1, M_2, N_j, N_nj):
    """fM_1,fM_2 (n,l), fN_j,fN_nj (k,n,l) for a vectorised objective.

    The reference calls a Python function row by row (varsens/saltelli.py:
    329-355) and always stores values 2-D (:313-321).
    """
    def two_d(v):
        v = np.asarray(v, dtype=np.float64)
        return v[..., None] if v.ndim == M_1.ndim - 1 else v
    fM_1 = two_d(rows_func(M_1))
    fM_2 = two_d(rows_func(M_2))
    fN_j = rows_func(N_j)
    fN_nj = rows_func(N_nj)
    if fN_j.ndim == 2:
        fN_j = fN_j[..., None]
        fN_nj = fN_nj[..., None]
    return fM_1, fM_2, np.asarray(fN_j, dtype=np.float64), np.asarray(fN_nj, dtype=np.float64)


def evaluate_callable(func, M_1, M_2, N_j, N_nj):
    """Row-by-row evaluation of an opaque callable (varsens/saltelli.py:308-355)."""
    n, k = M_1.shape
    test = func(M_1[0])
    try:
        l = len(test)
    except TypeError:
        l = 1
    fM_1 = np.zeros((n, l)); fM_2 = np.zeros((n, l))
    fN_j = np.zeros((k, n, l)); fN_nj = np.zeros((k, n, l))
    for r in range(n):
        fM_1[r] = func(M_1[r])
        fM_2[r] = func(M_2[r])
    for i in range(k):
        for r in range(n):
            fN_j[i][r] = func(N_j[i][r])
            fN_nj[i][r] = func(N_nj[i][r])
    return fM_1, fM_2, fN_j, fN_nj


def flat_objective(fM_1, fM_2, fN_j, fN_nj):
    """Objective.flat order (varsens/saltelli.py:357-391)."""
    l = fM_1.shape[1]
    return np.concatenate([fM_1, fM_2, fN_j.reshape(-1, l), fN_nj.reshape(-1, l)], axis=0)


def nan_row_mask(vals, k, n):
    """True for rows the reference deletes (varsens/saltelli.py:474-486).

    A row r goes when any of the 2k+2 blocks holds a NaN at r in the FIRST
    output column (:482).
    """
    v = np.asarray(vals, dtype=np.float64)
    if v.ndim == 1:
        v = v[:, None]
    first = v[:, 0].reshape(2 * k + 2, n)
    return np.isnan(first).any(axis=0)


def load_objective(vals, k, n, scaling=1.0):
    """Objective.load on an in-memory (2n(k+1), l) array (varsens/saltelli.py:449-492)."""
    x = np.asarray(vals, dtype=np.float64)
    if len(x) != 2 * n * (1 + k):
        raise Exception("Loaded objective has length " + str(len(x)) +
                        ". Must have length %d." % (2 * n * (1 + k)))
    l = x.shape[1]
    x = x / scaling
    fM_1 = x[0:n]
    fM_2 = x[n:2 * n]
    fN_j = x[2 * n:(2 + k) * n].reshape(k, n, l)
    fN_nj = x[(2 + k) * n:].reshape(k, n, l)
    drop = nan_row_mask(x, k, n)
    keep = ~drop
    return fM_1[keep], fM_2[keep], fN_j[:, keep], fN_nj[:, keep], int(drop.sum())


# ----------------------------------------------------------------------------
# a8-a12: compute_varsens
# ----------------------------------------------------------------------------


def compute_varsens(fM_1, fM_2, fN_j, fN_nj, n, second_order=True):
    """Reference arithmetic, operation for operation (varsens/saltelli.py:572-622).

    ``n`` is the REQUESTED n: divisors do not shrink when NaN rows were trimmed
    (:576 is commented out).  Returns a dict with the reference's attribute names.
    """
    k = fN_j.shape[0]
    out = {}
    E_2 = sum(fM_1 * fM_2) / n                                              # :577
    var_y = np.var(np.concatenate((fM_1, fM_2), axis=0), axis=0, ddof=1)    # :583
    U_j = np.sum(fM_1 * fN_j, axis=1) / (n - 1)                             # :591
    U_j += np.sum(fM_2 * fN_nj, axis=1) / (n - 1)                           # :592
    U_j /= 2.0                                                              # :593
    U_nj = np.sum(fM_1 * fN_nj, axis=1) / (n - 1)                           # :594
    U_nj += np.sum(fM_2 * fN_j, axis=1) / (n - 1)                           # :595
    U_nj /= 2.0                                                             # :596
    sens = np.zeros([k, U_j.shape[1]])
    sens_t = np.zeros([k, U_j.shape[1]])
    for j in range(k):
        sens[j] = (U_j[j] - E_2) / var_y                                    # :608
        sens_t[j] = 1.0 - ((U_nj[j] - E_2) / var_y)                         # :609
    out.update(E_2=E_2, var_y=var_y, U_j=U_j, U_nj=U_nj, sens=sens, sens_t=sens_t)
    if second_order:
        s2 = np.tensordot(fN_nj, fN_j, axes=([1], [1]))                     # :612
        s2 += np.tensordot(fN_j, fN_nj, axes=([1], [1]))                    # :613
        s2 /= 2.0 * (n - 1)
        s2 -= E_2
        s2 /= var_y
        s2n = np.tensordot(fN_nj, fN_nj, axes=([1], [1]))                   # :618
        s2n += np.tensordot(fN_j, fN_j, axes=([1], [1]))                    # :619
        s2n /= 2.0 * (n - 1)
        s2n -= E_2
        s2n /= var_y
        out.update(sens_2=s2, sens_2n=s2n)
    return out


def exact_accumulators(fM_1, fM_2, fN_j, fN_nj):
    """Correctly rounded raw sums the estimators are built from (math.fsum).

    Each product is rounded once to fp64 (as numpy does), the sums over rows are
    exact.  (The kernels accumulate with fma, i.e. unrounded products; the two
    differ by ~eps/sqrt(n) relative, far below every tolerance used.)  Names
    match the HIP accumulator record:
      sab (l,), sa (l,), sb (l,), m2 (l,) = sum (x-mean)^2 over [a;b],
      ac, bd, ad, bc each (k,l).
    """
    m, l = fM_1.shape
    k = fN_j.shape[0]

    def fs(v):
        return math.fsum(v.tolist())

    acc = dict(rows=m)
    acc["sab"] = np.array([fs(fM_1[:, o] * fM_2[:, o]) for o in range(l)])
    acc["sa"] = np.array([fs(fM_1[:, o]) for o in range(l)])
    acc["sb"] = np.array([fs(fM_2[:, o]) for o in range(l)])
    mean = (acc["sa"] + acc["sb"]) / (2.0 * m) if m else np.zeros(l)
    acc["mean"] = mean
    acc["m2"] = np.array([fs((fM_1[:, o] - mean[o]) ** 2) + fs((fM_2[:, o] - mean[o]) ** 2)
                          for o in range(l)])
    for name, x, y in (("ac", fM_1, fN_j), ("bd", fM_2, fN_nj),
                       ("ad", fM_1, fN_nj), ("bc", fM_2, fN_j)):
        acc[name] = np.array([[fs(x[:, o] * y[j, :, o]) for o in range(l)] for j in range(k)])
    return acc


def indices_from_accumulators(acc, n):
    """Finish E_2 .. sens_t from raw sums with the reference's divisors
    (varsens/saltelli.py:577,583,591-596,608-609)."""
    m = acc["rows"]
    E_2 = acc["sab"] / n
    var_y = acc["m2"] / (2 * m - 1)
    U_j = (acc["ac"] / (n - 1) + acc["bd"] / (n - 1)) / 2.0
    U_nj = (acc["ad"] / (n - 1) + acc["bc"] / (n - 1)) / 2.0
    sens = (U_j - E_2) / var_y
    sens_t = 1.0 - ((U_nj - E_2) / var_y)
    return dict(E_2=E_2, var_y=var_y, U_j=U_j, U_nj=U_nj, sens=sens, sens_t=sens_t)


def exact_second_order(fN_j, fN_nj, E_2, var_y, n):
    """sens_2 / sens_2n from exactly summed Gram entries (:612-622)."""
    k, m, l = fN_j.shape
    C = fN_j.transpose(0, 2, 1).reshape(k * l, m)
    D = fN_nj.transpose(0, 2, 1).reshape(k * l, m)

    def gram(X, Y):
        return np.array([[math.fsum((X[i] * Y[j]).tolist()) for j in range(k * l)]
                         for i in range(k * l)]).reshape(k, l, k, l)
    dc = gram(D, C)
    cd = gram(C, D)
    dd = gram(D, D)
    cc = gram(C, C)
    s2 = ((dc + cd) / (2.0 * (n - 1)) - E_2) / var_y
    s2n = ((dd + cc) / (2.0 * (n - 1)) - E_2) / var_y
    return s2, s2n


# ----------------------------------------------------------------------------
# analytic truths used by the tests (varsens/tests/test_g_function.py:21-50)
# ----------------------------------------------------------------------------


def g_partial_variances(a):
    return 1.0 / (3.0 * (np.asarray(a, dtype=np.float64) + 1.0) ** 2)


def product_form_indices(V):
    """S_i and ST_i for f = prod_i g_i(x_i) with E g_i = 1, Var g_i = V_i."""
    V = np.asarray(V, dtype=np.float64)
    total = np.prod(1.0 + V) - 1.0
    S = V / total
    ST = V * (np.prod(1.0 + V) / (1.0 + V)) / total
    return total, S, ST


def ishigami_truth(a=7.0, b=0.1):
    pi = math.pi
    V = a * a / 8 + b * pi ** 4 / 5 + b * b * pi ** 8 / 18 + 0.5
    V1 = 0.5 * (1 + b * pi ** 4 / 5) ** 2
    V2 = a * a / 8
    V13 = 8 * b * b * pi ** 8 / 225
    S = np.array([V1, V2, 0.0]) / V
    ST = np.array([V1 + V13, V2, V13]) / V
    return V, S, ST
