#!/usr/bin/env python3
"""Accuracy profile of the first-order estimator on random Sobol g-functions, on the GPU.

Restates the study behind the reference's only published numbers
(varsens/profile/parameters.py:51-83, error-profile-dim*.csv): for a Halton design of n draws in k
dimensions, draw b random models a_i ~ U(0,99), estimate S_i * var_y for each and record
sum_i (V_i - S_i*var_y)^2.  Columns: n, mean, sd, lower CI, upper CI, max (as in the reference's CSVs),
then the median and median / k.

What the published curve is (profiles/r02/halton_variants_k6.txt, tools/halton_variants.py): with
a_i ~ U(0,99) the model is f = 1 + O(1e-2), so S_i*var_y = U_j - E_2 is dominated by the estimator's
own bias n/(n-1) - 1 = 1/(n-1) (U_j divides by n-1, E_2 by n, varsens/saltelli.py:577,591): the error is
k/(n-1)^2 whatever the points are -- every Halton variant tried gives a MEDIAN of k/n^2 to 1-3 %.  The
published means equal 1/n^2 to 2 % from n = 80 on (and the dim-6 and dim-12 files are byte-identical):
parameters.py:74 holds a commented-out `error /= k`; with it active the published numbers are exactly this.
So the column to read against the published mean is median / k.  (Means are pulled up by the rare model
with some a_i < 1, whose true Monte Carlo error dominates.)

    python examples/accuracy_profile.py --k 6 --b 30
    python examples/accuracy_profile.py --k 24 --b 30 --nmax 1310720     # far beyond the reference's 20480
    python examples/accuracy_profile.py --all --b 30 --out profiles/r03  # dims 6, 24, 48, 96, 192 beside the published CSVs

Every model is evaluated and reduced on the device (fused evaluate+reduce kernel); the design is built
once per n, as in the reference.  Per model the device's numbers equal the reference arithmetic's on the
same model to 1e-10 (tests/test_gpu_api.py::test_accuracy_profile_errors_equal_the_oracles); what differs
from the published files is WHICH models were drawn (the reference used Python's global `random`, unseeded).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import varsens_amd as vs  # noqa: E402

# mean column of varsens/profile/error-profile-dim6.csv (n -> published mean), for side-by-side reading
PUBLISHED_DIM6 = {5: 5.982e-2, 10: 1.538e-2, 20: 3.314e-3, 40: 6.941e-4, 80: 1.580e-4, 160: 3.999e-5, 320: 1.230e-5,
                  640: 2.431e-6, 1280: 7.431e-7, 2560: 1.517e-7, 5120: 3.862e-8, 10240: 9.736e-9, 20480: 2.410e-9}


def published(k):
    """{n: (mean, sd, lci, uci, max)} of varsens/profile/error-profile-dim<k>.csv (fixture: numbers only), or {}."""
    path = os.path.join(ROOT, "tests", "golden", "published_error_profiles.json")
    if not os.path.isfile(path):
        return {}
    rows = json.load(open(path))["dims"].get(str(k), [])
    return {int(r[0]): tuple(r[1:]) for r in rows}


def g_truth(a):
    return 1.0 / (3.0 * (np.asarray(a) + 1.0) ** 2)


def profile_row(k, n, b, rng, second_order=False):
    """`second_order`: run the reference's own call -- Varsens(...) always computes sens_2 / sens_2n too
    (varsens/saltelli.py:612-622, called from varsens/profile/parameters.py:63) -- instead of the first order alone;
    the error metric is the same (it reads `sens` only), the work is not: at k = 192 every model then goes through the
    evaluation kernel and the Gram kernels for 192 index rows per half."""
    s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
    errs = []
    for _ in range(b):
        a = rng.uniform(0.0, 99.0, size=k)
        v = vs.Varsens(vs.objectives.GFunction(a), sample=s, verbose=False, second_order=second_order)
        estimate = v.sens[:, 0] * v.var_y[0]
        errs.append(float(np.sum((g_truth(a) - estimate) ** 2)))
        if second_order:
            assert v.sens_2.shape == (k, 1, k, 1) and np.isfinite(v.sens_2).all()
    errs = np.array(errs)
    mu, sd = errs.mean(), errs.std()
    half = 0.0
    if b > 1:
        from scipy.stats import t
        half = sd / np.sqrt(b) * t.isf(0.025, b - 1)
    return n, mu, sd, mu - half, mu + half, errs.max(), float(np.median(errs)), float(np.median(errs)) / k


def model_error(k, a, sample):
    """sum_i (V_i - S_i var_y)^2 of one model on the device (varsens/profile/parameters.py:51-76)."""
    v = vs.Varsens(vs.objectives.GFunction(a), sample=sample, verbose=False, second_order=False)
    estimate = v.sens[:, 0] * v.var_y[0]
    return float(np.sum((g_truth(a) - estimate) ** 2)), estimate


def run_dim(k, b, nmax, seed, out=None, second_order=False):
    import time
    rng = np.random.RandomState(seed)
    pub = published(k)
    lines = ["# k=%d b=%d seed=%d%s; device: n, mean, sd, lci, uci, max, median, median/k | published "
             "(varsens/profile/error-profile-dim%d.csv): mean, sd, max | 1/n^2 | wall ms per model"
             % (k, b, seed, " second_order=True (the reference's default call)" if second_order else "", k)]
    n = 5
    while n <= nmax:
        t0 = time.perf_counter()
        row = profile_row(k, n, b, rng, second_order)
        per_model_ms = 1e3 * (time.perf_counter() - t0) / b
        line = "%d,%.6e,%.6e,%.6e,%.6e,%.6e,%.6e,%.6e" % row
        if n in pub:
            line += " | %.6e,%.6e,%.6e" % (pub[n][0], pub[n][1], pub[n][4])
        line += " | %.6e | %.3f" % (1.0 / n ** 2, per_model_ms)
        lines.append(line)
        print(line, flush=True)
        n *= 2
    if out:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "accuracy_profile_k%d_b%d%s.csv" % (k, b, "_second_order" if second_order else "")), "w") as fh:
            fh.write("\n".join(lines) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--b", type=int, default=30)
    ap.add_argument("--nmax", type=int, default=20480)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--all", action="store_true", help="the five published dimensionalities 6, 24, 48, 96, 192")
    ap.add_argument("--dims", default=None, help="comma-separated dimensionalities, e.g. 192")
    ap.add_argument("--second-order", action="store_true",
                    help="the reference's default call: sens_2 / sens_2n computed for every model as well")
    ap.add_argument("--out", default=None, help="directory for accuracy_profile_k<k>_b<b>.csv")
    args = ap.parse_args()
    dims = [int(x) for x in args.dims.split(",")] if args.dims else ((6, 24, 48, 96, 192) if args.all else (args.k,))
    for k in dims:
        print("# k=%d b=%d  columns: n, mean, sd, lci, uci, max, median, median/k | published mean, sd, max | 1/n^2 "
              "| wall ms per model" % (k, args.b))
        run_dim(k, args.b, args.nmax, args.seed, args.out, args.second_order)


if __name__ == "__main__":
    main()
