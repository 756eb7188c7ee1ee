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


def horner_numerator(i, b, ndig):
    """vs_radical_inverse_pair_horner's numerator, operation for operation (every fp64 instruction an exactly
    rounded one over Fractions): three instructions per digit step, then -(b^2 - 1) P + i top."""
    inv_b = inv_b_up(b)
    bd, top = float(b), float(b ** (ndig - 1))
    j, p = float(i), 0.0
    for _ in range(ndig - 1):
        j = float(math.floor(float(Fraction(j) * Fraction(inv_b))))   # v_mul_f64 (rounded), v_floor_f64
        p = fma(p, bd, j)
    m = fma(-bd, bd, 1.0)
    return fma(m, p, float(Fraction(float(i)) * Fraction(top)))


def test_horner_form_gives_the_digit_reversed_integer():
    """sum_t d_t b^(T-t) = j_0 b^T - (b^2 - 1) sum_{t>=1} j_t b^(T-t) with j_t = i // b^t: the numerator the device forms
    without forming a digit is the exact digit-reversed integer whenever (largest index) * b^(ndig-1) < 2^53 -- the
    host's condition for VS_BASE_HORNER -- at the edges of that condition and of the digit counts."""
    rng = np.random.RandomState(2)
    bases = [int(b) for b in orc.first_primes(1000)]
    for b in bases[1:14] + bases[30:34] + bases[125:129] + bases[-2:]:
        for max_index in (5, b - 1, b, b * b, 641 + (1 << 23), 2561 + (1 << 25), 94_900_000, (1 << 28) + 7, (1 << 32) - 1):
            ndig, v = 0, max_index
            while v:
                ndig += 1
                v //= b
            ok = max_index * b ** (ndig - 1) < 2 ** 53
            idx = {1, 2, b - 1, b, b + 1, min(b * b, max_index), max_index, max_index - 1, b ** (ndig - 1), b ** (ndig - 1) - 1}
            idx.update(int(x) for x in rng.randint(1, max_index + 1, 60))
            for i in idx:
                if not 1 <= i <= max_index:
                    continue
                num, j = 0, i
                for _ in range(ndig):
                    num = num * b + j % b
                    j //= b
                got = horner_numerator(i, b, ndig)
                if ok:
                    assert got == float(num) and Fraction(got) == num, (i, b, ndig)
    # the condition is the one that matters: beyond it the product i * top is no longer exact for some index
    b, max_index = 3, (1 << 32) - 1
    ndig = 21
    assert 3 ** 20 <= max_index < 3 ** 21 and max_index * 3 ** 20 >= 2 ** 53
    wrong = 0
    for i in range(max_index - 2000, max_index):
        num, j = 0, i
        for _ in range(ndig):
            num = num * b + j % b
            j //= b
        wrong += Fraction(horner_numerator(i, b, ndig)) != num
    assert wrong > 0


def two_word_reciprocal(den):
    """the host's (inv_den, inv_lo) of make_base (vs_generate.hip): RN(1/den) and RN of what is left"""
    hi = 1.0 / float(den)
    r = fma(-float(den), hi, 1.0)            # exact: the residual of a correctly rounded reciprocal fits a double
    assert Fraction(r) == 1 - Fraction(den) * Fraction(hi)
    return hi, r / float(den)


def quotient_dd(n, hi, lo):
    """vs_halton_quotients_dd: one rounded product and one fma"""
    return fma(float(n), hi, float(Fraction(float(n)) * Fraction(lo)))


def test_two_instruction_quotient_is_the_correctly_rounded_one():
    """x = fma(n, inv_den, RN(n inv_lo)) equals RN(n / den) for every numerator below a denominator b^ndig <= 2^49 (the
    host's condition for VS_BASE_HORNER): random numerators, the ends of the range, and the numerators CLOSEST to a
    rounding boundary -- for a boundary m (odd multiple of half an ulp) the integers next to den * m."""
    rng = np.random.RandomState(3)
    bases = [int(b) for b in orc.first_primes(1000)]
    checked = 0
    for b in bases[1:16] + bases[60:64] + bases[125:129] + bases[-3:]:
        for ndig in range(3, 40):
            den = b ** ndig
            if den > 2 ** 49:
                break
            if den < 2 ** 20 and ndig > 3:
                continue
            hi, lo = two_word_reciprocal(den)
            assert lo == float(Fraction(1, den) - Fraction(hi))           # RN(1/den - inv_den)
            ns = {1, 2, den - 1, den - 2, den // 2, den // 2 + 1, den // 3, (den * 2) // 3}
            ns.update(int(x) for x in rng.randint(1, min(den, 2 ** 62), 40, dtype=np.int64) % den)
            # numerators next to rounding boundaries of quotients in [1/2, 1) and [1/4, 1/2): boundary = (2 j + 1) 2^-54, 2^-55
            for e in (54, 55, 60):
                for _ in range(12):
                    j = int(rng.randint(2 ** 51, 2 ** 52, dtype=np.int64))
                    mid = Fraction(2 * j + 1, 2 ** e)
                    if e == 54 and not (Fraction(1, 2) <= mid < 1):
                        continue
                    t = den * mid
                    ns.update((int(t), int(t) + 1))
            for n in ns:
                if 0 < n < den:
                    assert quotient_dd(n, hi, lo) == float(Fraction(n, den)), (n, b, ndig)
                    checked += 1
    assert checked > 5000
