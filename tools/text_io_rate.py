"""The text wire format (Sample.export / .load, Objective.export / .load; varsens/saltelli.py:173-250, 393-472) timed
against numpy.savetxt / numpy.loadtxt on the same arrays, on the box's host cores; files in a scratch directory under
/dev/shm (memory: the rates are the formatting and parsing, not a disk's).

  python tools/text_io_rate.py [k] [log2 n]        (default 32, 14: 2n(k+1) = 1.08 M rows, 34.6 M sample values)
"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import varsens_amd as vs
from varsens_amd import _native


def timed(fn):
    t0 = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t0


def numpy_rows(sample, count):
    return np.ascontiguousarray(sample._rows(0, count))


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 14)
    rows = 2 * n * (k + 1)
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        print("k = %d, n = %d: %d rows, %.1f M sample values; host threads: %d" %
              (k, n, rows, rows * k / 1e6, os.cpu_count()))
        s = vs.Sample(k, n, vs.scale.Identity(), verbose=False)
        _, t = timed(lambda: s.export(outdir=d, prefix="sample"))
        size = os.path.getsize(os.path.join(d, "sample.txt"))
        print("Sample.export            %7.2f s  %6.1f ns/value  (%.0f MB of text, %.2f GB/s)" %
              (t, 1e9 * t / (rows * k), size / 1e6, size / t / 1e9))
        s2, t = timed(lambda: vs.Sample(k, n, vs.scale.Identity(), verbose=False, loadFile=os.path.join(d, "sample.txt")))
        print("Sample.load              %7.2f s  %6.1f ns/value" % (t, 1e9 * t / (rows * k)))

        # numpy on a bounded slice of the same rows (one core; the whole file would take minutes)
        part = min(rows, 1 << 15)
        x = numpy_rows(s, part)
        _, t = timed(lambda: np.savetxt(os.path.join(d, "np.txt"), x, delimiter="\t"))
        print("numpy.savetxt (%d rows)  %7.2f s  %6.1f ns/value" % (part, t, 1e9 * t / x.size))
        y, t = timed(lambda: np.loadtxt(os.path.join(d, "np.txt"), delimiter="\t", ndmin=2))
        print("numpy.loadtxt (%d rows)  %7.2f s  %6.1f ns/value" % (part, t, 1e9 * t / x.size))
        _native.text_write(os.path.join(d, "vs.txt"), x, "\t")
        same = open(os.path.join(d, "np.txt"), "rb").read() == open(os.path.join(d, "vs.txt"), "rb").read()
        print("bytes identical to numpy.savetxt: %s; values identical to numpy.loadtxt: %s" %
              (same, bool(np.array_equal(_native.text_read(os.path.join(d, "np.txt"), "\t"), y))))
        os.remove(os.path.join(d, "sample.txt"))

        # objective values of a 3-output model: export, load, analyse
        a = [0, .5, 3, 9, 99, 99] + [99] * (k - 6)
        o = vs.Objective(k, n, s, vs.objectives.GFunction([a, a[::-1], a]), verbose=False)
        _, t = timed(lambda: o.export(outdir=d, prefix="objective"))
        print("Objective.export (l=3)   %7.2f s  %6.1f ns/value" % (t, 1e9 * t / (rows * 3)))
        o2, t = timed(lambda: vs.Objective(k, n, verbose=False, loadFile=os.path.join(d, "objective.txt")))
        print("Objective.load (l=3)     %7.2f s  %6.1f ns/value" % (t, 1e9 * t / (rows * 3)))
        v1 = vs.Varsens(o, verbose=False)
        v2 = vs.Varsens(o2, verbose=False)
        print("sens from the loaded file equals sens from the device values bit for bit: %s" %
              bool(np.array_equal(v1.sens, v2.sens) and np.array_equal(v1.sens_t, v2.sens_t)))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
