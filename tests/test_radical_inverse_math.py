"""The two exactness claims behind the device Halton generator, checked on the CPU with exact
rational arithmetic (fractions.Fraction; float(Fraction) is the correctly rounded double, so an fma
is float(a*b + c) over Fractions):

1. q = floor(j * inv_b) with inv_b = RN(1/b) pushed up two ulps equals j // b for every j < 2^32
   (adversarial j: multiples of b, multiples minus one, the top of the range);
2. Markstein's two Newton steps on RN(1/den) return the correctly rounded quotient num/den.

These restate vs_radical_inverse_pair (varsens_amd/csrc/vs_device_gen.h) operation for operation.
"""
import math
from fractions import Fraction

import numpy as np

from oracle import saltelli_oracle as orc


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def inv_b_up(b):
    return math.nextafter(math.nextafter(1.0 / b, 2.0), 2.0)


def device_radical_inverse(i, b, ndig):
    inv_b = inv_b_up(b)
    den = float(b ** ndig)
    inv_den = 1.0 / den
    n, j = 0.0, float(i)
    for _ in range(ndig):
        q = math.floor(j * inv_b)
        n = fma(n, float(b), fma(-q, float(b), j))
        j = q
    q0 = n * inv_den
    r0 = fma(-q0, den, n)
    q1 = fma(r0, inv_den, q0)
    r1 = fma(-q1, den, n)
    return fma(r1, inv_den, q1)


def test_floor_of_inflated_reciprocal_is_exact_quotient():
    rng = np.random.RandomState(0)
    bases = [int(b) for b in orc.first_primes(1000)]
    for b in bases[:200] + bases[-50:]:
        inv = inv_b_up(b)
        top = (1 << 32) - 1
        js = set(int(x) for x in rng.randint(0, 1 << 32, 200, dtype=np.uint64))
        for m in list(rng.randint(1, top // b, 200)) + [1, 2, top // b]:
            js.update((int(m) * b, int(m) * b - 1, int(m) * b + 1))
        js.update((0, 1, b - 1, b, top, top - 1))
        for j in js:
            if 0 <= j <= top:
                assert math.floor(float(j) * inv) == j // b, (j, b)


def test_device_sequence_is_the_correctly_rounded_quotient():
    rng = np.random.RandomState(1)
    bases = [int(b) for b in orc.first_primes(128)]
    for b in bases[1:12] + bases[40:44] + bases[-3:]:
        for max_index in (641 + (1 << 23), 2561 + (1 << 26), (1 << 32) - 1):
            ndig = len(np.base_repr(max_index, 2)) and 0
            v = max_index
            while v:
                ndig += 1
                v //= b
            assert b ** ndig < 2 ** 53
            idx = [1, 2, b - 1, b, b + 1, b * b, max_index, max_index - 1] + [int(x) for x in rng.randint(1, max_index, 40)]
            for i in idx:
                # exact digit-reversed integer over b^ndig (padding digits included)
                num, j = 0, i
                for _ in range(ndig):
                    num = num * b + j % b
                    j //= b
                want = float(Fraction(num, b ** ndig))
                assert device_radical_inverse(i, b, ndig) == want, (i, b, ndig)
                # and the padded quotient is the un-padded one the oracle defines
                m, j = 0, i
                while j:
                    m += 1
                    j //= b
                assert want == float(Fraction(num // b ** (ndig - m), b ** m))
