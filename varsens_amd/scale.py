"""Maps from the unit cube to parameter ranges.

Two forms of each map:

* the reference's plain functions ``linear / power / percentage / magnitude``
  (varsens/scale.py:6,35,64,93) -- host numpy helpers a user calls inside a
  scaling lambda; same arithmetic, same argument names;
* device specs ``Linear / Power / Percentage / Magnitude / Identity`` --
  objects that can be passed as ``scaling`` to ``Sample``/``Varsens``.  They
  describe the map to the HIP generator kernel, which then applies it while it
  produces the points (no (n,k) block ever visits the host).  They are
  callable too, so code that applies ``scaling(block)`` by hand keeps working.

An opaque callable as ``scaling`` still works (reference behaviour): the unit
points are generated on the GPU, the callable runs on the host on the two
(n,k) halves exactly as varsens/saltelli.py:92,95 does, and the scaled halves
go back to the GPU.
"""
import numpy

from . import _native

__all__ = ['linear', 'power', 'percentage', 'magnitude',
           'Identity', 'Linear', 'Power', 'Percentage', 'Magnitude']


def linear(points, lower_bound, upper_bound):
    """[0,1] -> [lower, upper], affine (varsens/scale.py:33).

    >>> import numpy
    >>> linear(numpy.array([0.5]*3), numpy.array([-100, -10, 1000]), numpy.array([100, 20, 2000]))
    array([   0.,    5., 1500.])
    """
    width = upper_bound - lower_bound
    return points * width + lower_bound


def power(points, lower_bound, upper_bound):
    """[0,1] -> [lower, upper], geometric; bounds must be positive (varsens/scale.py:62).

    >>> import numpy
    >>> power(numpy.array([0.5]*2), numpy.array([10., 100.]), numpy.array([1000., 200.]))
    array([100.        , 141.42135624])
    """
    ratio = upper_bound / lower_bound
    return lower_bound * (ratio ** points)


def percentage(points, reference, percentage=50.0):
    """[0,1] -> reference +/- percentage % (varsens/scale.py:90-91).

    >>> import numpy
    >>> percentage(numpy.array([0.333]*3), numpy.array([1, 10, 1000]), 50.0)
    array([  0.833,   8.33 , 833.   ])
    """
    half_width = percentage * reference / 100.0
    return linear(points, reference - half_width, reference + half_width)


def magnitude(points, reference, orders=3.0, base=10.0):
    """[0,1] -> reference * base**(-orders..+orders), geometric (varsens/scale.py:121-122).

    >>> import numpy
    >>> magnitude(numpy.array([0.333]*3), numpy.array([1, 10, 1000]))
    array([ 0.09954054,  0.99540542, 99.54054174])
    """
    span = base ** orders
    return power(points, reference / span, reference * span)


class DeviceScale(object):
    """A scaling the generator kernel applies itself.  ``device_spec(k)`` returns
    (mode, lower[k], upper[k]) for vs_design_desc."""
    mode = _native.SCALE_IDENTITY

    def device_spec(self, k):
        return self.mode, None, None

    @staticmethod
    def _bounds(v, k, name):
        a = numpy.asarray(v, dtype=numpy.float64)
        if a.ndim == 0:
            a = numpy.full(k, float(a))
        if a.shape != (k,):
            raise Exception("%s must be a scalar or have %d entries, got shape %s" % (name, k, a.shape))
        return a


class Identity(DeviceScale):
    """x -> x (what the reference falls back to when scaling is None, varsens/saltelli.py:86-88)."""

    def __call__(self, points):
        return points


class Linear(DeviceScale):
    mode = _native.SCALE_LINEAR

    def __init__(self, lower_bound, upper_bound):
        self.lower_bound, self.upper_bound = lower_bound, upper_bound

    def __call__(self, points):
        return linear(points, numpy.asarray(self.lower_bound, dtype=float), numpy.asarray(self.upper_bound, dtype=float))

    def device_spec(self, k):
        return self.mode, self._bounds(self.lower_bound, k, "lower_bound"), self._bounds(self.upper_bound, k, "upper_bound")


class Power(Linear):
    mode = _native.SCALE_POWER

    def __call__(self, points):
        return power(points, numpy.asarray(self.lower_bound, dtype=float), numpy.asarray(self.upper_bound, dtype=float))

    def device_spec(self, k):
        mode, lo, hi = Linear.device_spec(self, k)
        if not (numpy.all(lo > 0) and numpy.all(hi > 0)):
            raise Exception("power scaling needs positive bounds")
        return mode, lo, hi


class Percentage(Linear):
    def __init__(self, reference, percentage=50.0):
        ref = numpy.asarray(reference, dtype=numpy.float64)
        half_width = percentage * ref / 100.0          # varsens/scale.py:90
        Linear.__init__(self, ref - half_width, ref + half_width)


class Magnitude(Power):
    def __init__(self, reference, orders=3.0, base=10.0):
        ref = numpy.asarray(reference, dtype=numpy.float64)
        span = base ** orders                          # varsens/scale.py:121
        Power.__init__(self, ref / span, ref * span)
