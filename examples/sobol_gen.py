#!/usr/bin/env python3
"""Sobol sample files in the shape of the reference's quantlib/sobolGen.cpp, from the engine's device generator.

    python3 examples/sobol_gen.py dimensions samples [seed] [--skip 4096] [--precision N] [--no-header] > sobol.csv

The reference's tool (quantlib/sobolGen.cpp:11-63) prints `samples` points of a `dimensions`-dimensional Sobol sequence
after `skipTo(4096)` (:50), one point per line, coordinates separated by commas (:52-63); the file is meant to be read
back as an UNSCALED (2n, k) sample: ``Sample(k, n, scaling, loadFile='sobol.csv', delimiter=',')``
(varsens/saltelli.py:43-46, 224-227), i.e. samples = 2n.  This is the same command line over ``vs_sobol_fill``
(include/varsens_hip.h).

What is the same: the command line, the skip of 4096 points, the CSV shape, the route through ``Sample(loadFile=...)``.
What is NOT: the direction integers.  QuantLib is asked for ``SobolRsg::SobolLevitanLemieux`` (:47); those integers are
not available here (no QuantLib in this image, no network), so the points are the Joe-Kuo sequence -- bit for bit
``scipy.stats.qmc.Sobol(scramble=False)`` -- and value parity with the reference's own files stays UNPINNED
(DESIGN.md section 6).  The optional third argument, QuantLib's seed (:34), is accepted and ignored: an unscrambled Sobol
sequence has no seed.  Coordinates are printed with the shortest text that reads back to the same double (the
reference prints C++'s default six significant digits; ``--precision 6`` does that), so a file read back through
``Sample(loadFile=...)`` gives the in-memory ``generator='sobol'`` design bit for bit
(tests/test_gpu_api.py::test_sobol_file_round_trip_equals_the_in_memory_design).
"""
import argparse
import os
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def sobol_points(dimensions, samples, skip=4096, context=None):
    """(samples, dimensions) points of indices skip .. skip + samples - 1, generated on the GPU."""
    from varsens_amd.saltelli import default_context, sobol_directions
    ctx = context or default_context()
    buf = ctx.sobol_fill(dimensions, sobol_directions(dimensions), skip, samples)
    try:
        return ctx.download(buf, (samples, dimensions))
    finally:
        buf.free()


def write_csv(points, out, precision=None, header=True, skip=4096):
    if header:    # numpy.loadtxt -- the reference's reader (varsens/saltelli.py:215) -- skips '#' lines
        out.write("# Sobol points %d .. %d, %d dimensions, Joe-Kuo direction integers (scipy.stats.qmc.Sobol, unscrambled);\n"
                  "# NOT the Levitan-Lemieux integers of the reference's quantlib/sobolGen.cpp: same shape, other points\n"
                  % (skip, skip + len(points) - 1, points.shape[1]))
    fmt = repr if precision is None else (lambda x: "%.*g" % (precision, x))
    for row in points:
        out.write(",".join(fmt(float(x)) for x in row))
        out.write("\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description="Sobol points as CSV rows (cf. quantlib/sobolGen.cpp of the reference)",
                                 usage="sobol_gen.py dimensions samples [seed] [--skip 4096] [--precision N] [--no-header]")
    ap.add_argument("dimensions", type=int)
    ap.add_argument("samples", type=int)
    ap.add_argument("seed", type=int, nargs="?", default=64, help="accepted for command-line compatibility; ignored")
    ap.add_argument("--skip", type=int, default=4096, help="index of the first point (quantlib/sobolGen.cpp:50)")
    ap.add_argument("--precision", type=int, default=None,
                    help="significant digits (default: shortest round-trip text; 6 = the reference's iostream default)")
    ap.add_argument("--no-header", action="store_true")
    args = ap.parse_args(argv)
    if args.dimensions < 1 or args.samples < 1:
        ap.error("dimensions and samples must be positive")     # (the reference: "Invalid dimension" / "Invalid samples")
    write_csv(sobol_points(args.dimensions, args.samples, args.skip), sys.stdout, args.precision,
              not args.no_header, args.skip)
    return 0


if __name__ == "__main__":
    sys.exit(main())
