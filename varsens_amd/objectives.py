"""Objective functions the GPU evaluates itself (vs_objective_create).

The reference takes any Python callable and calls it once per sample row
(varsens/saltelli.py:308,329-355).  That still works here (the rows come from
the GPU, the callable runs on the host), but an instance of one of these
classes is evaluated inside the fused generate->evaluate kernel instead.
"""
import numpy

from . import _native


class DeviceObjective(object):
    kind = None

    def outputs(self, k):
        return 1

    def params(self, k):
        return numpy.zeros(0)

    def _table(self, v, k, name):
        a = numpy.atleast_2d(numpy.asarray(v, dtype=numpy.float64))
        if a.ndim != 2 or a.shape[1] != k:
            raise Exception("%s must have k=%d entries per output, got shape %s" % (name, k, a.shape))
        return a


class GFunction(DeviceObjective):
    """Sobol g-function prod_i (|4 x_i - 2| + a_i)/(1 + a_i) (varsens/tests/test_g_function.py:9-14).
    ``a`` of shape (k,) gives one output, (l,k) gives l outputs."""
    kind = _native.OBJ_G_FUNCTION

    def __init__(self, a):
        self.a = a

    def outputs(self, k):
        return self._table(self.a, k, "a").shape[0]

    def params(self, k):
        return self._table(self.a, k, "a")


class Linear(DeviceObjective):
    """sum_i c_i x_i."""
    kind = _native.OBJ_LINEAR

    def __init__(self, c):
        self.c = c

    def outputs(self, k):
        return self._table(self.c, k, "c").shape[0]

    def params(self, k):
        return self._table(self.c, k, "c")


class Polynomial(DeviceObjective):
    """prod_i (1 + beta_i (x_i - 1/2))."""
    kind = _native.OBJ_POLYNOMIAL

    def __init__(self, beta):
        self.beta = beta

    def outputs(self, k):
        return self._table(self.beta, k, "beta").shape[0]

    def params(self, k):
        return self._table(self.beta, k, "beta")


class Ishigami(DeviceObjective):
    """sin x1 + a sin^2 x2 + b x3^4 sin x1 (k = 3)."""
    kind = _native.OBJ_ISHIGAMI

    def __init__(self, a=7.0, b=0.1):
        self.a, self.b = float(a), float(b)

    def params(self, k):
        if k != 3:
            raise Exception("the Ishigami function has k=3 parameters")
        return numpy.array([self.a, self.b])


class Invert(DeviceObjective):
    """x -> 1 - x, k outputs (varsens/tests/test_objective.py:5)."""
    kind = _native.OBJ_INVERT

    def outputs(self, k):
        return k


class Expression(DeviceObjective):
    """A user objective written as a HIP C++ body, compiled at run time (hiprtc) into the fused
    generate->evaluate kernel -- the device counterpart of an arbitrary Python callable.

    ``body`` sees ``x[c]`` (scaled coordinate c of the point), ``k``, ``p[...]`` (``params``) and
    writes ``out[0..outputs)``::

        Expression("out[0] = sin(x[0]) + p[0]*sin(x[1])*sin(x[1]) + p[1]*pow(x[2],4.0)*sin(x[0]);",
                   params=[7.0, 0.1])
    """
    kind = _native.OBJ_SOURCE

    def __init__(self, body, outputs=1, params=()):
        self.body = str(body)
        self._outputs = int(outputs)
        self._params = numpy.asarray(params, dtype=numpy.float64).ravel()

    def outputs(self, k):
        return self._outputs

    def params(self, k):
        return self._params


class SeparableExpression(DeviceObjective):
    """A user objective of separable shape, f(x) = t_0(x_0) op t_1(x_1) op ... with op = '*' or '+',
    given as the body of ``double term(double x, int c, const double* p)``.  Compiled at run time
    into the chain-scheme kernel of the built-in separable objectives, so it runs at their speed::

        SeparableExpression("return (fabs(4.0*x - 2.0) + p[c]) / (1.0 + p[c]);", fold="prod", params=a)
    """
    kind = _native.OBJ_SOURCE

    def __init__(self, term_body, fold="prod", params=()):
        if fold not in ("prod", "sum"):
            raise Exception("fold must be 'prod' or 'sum'")
        self.term_body = str(term_body)
        self.fold_sum = fold == "sum"
        self._params = numpy.asarray(params, dtype=numpy.float64).ravel()

    def params(self, k):
        return self._params
