#!/usr/bin/env python3
"""Coefficients of the double-precision momentum draw (PBBI_DRAW_F64; include/pbbi.h "RNG contract"):
Taylor coefficients of sin / cos((pi/2) y) in y, rounded to double from 60-digit decimals, printed as C hex
literals -- pasted into csrc/pbbi_rng.h (device) and oracle/pbbi_oracle.c (host restatement).  Also checks a
Python prototype of the transform's pieces (same operation order, exact fma) against math.log / sin / cos."""
from decimal import Decimal, getcontext
from fractions import Fraction
import math
import random
import struct

getcontext().prec = 60
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494")
h = PI / 2


def fact(n):
    r = Decimal(1)
    for i in range(2, n + 1):
        r *= i
    return r


S = [float((-1) ** k * h ** (2 * k + 1) / fact(2 * k + 1)) for k in range(9)]
C = [float((-1) ** k * h ** (2 * k) / fact(2 * k)) for k in range(10)]
LG = [0x3FE5555555555593, 0x3FD999999997FA04, 0x3FD2492494229359, 0x3FCC71C51D8E78AF, 0x3FC7466496CB03DE,
      0x3FC39A09D078C69F, 0x3FC2F112DF3E5244]
Lg = [struct.unpack("<d", struct.pack("<Q", v))[0] for v in LG]
ln2_hi = struct.unpack("<d", struct.pack("<Q", 0x3FE62E42FEE00000))[0]
ln2_lo = struct.unpack("<d", struct.pack("<Q", 0x3DEA39EF35793C76))[0]


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def log_u(u):
    bits = struct.unpack("<Q", struct.pack("<d", u))[0]
    e = (bits >> 52) - 1023
    mant = bits & ((1 << 52) - 1)
    if mant > 0x6A09E667F3BCC:
        e += 1
        mbits = mant | (1022 << 52)
    else:
        mbits = mant | (1023 << 52)
    m = struct.unpack("<d", struct.pack("<Q", mbits))[0]
    f = m - 1.0
    s = f / (2.0 + f)
    z = s * s
    w = z * z
    t1 = w * fma(w, fma(w, Lg[5], Lg[3]), Lg[1])
    t2 = z * fma(w, fma(w, fma(w, Lg[6], Lg[4]), Lg[2]), Lg[0])
    R = t2 + t1
    hfsq = 0.5 * f * f
    dk = float(e)
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f)


def sincos_quarter(y):
    z = y * y
    s = S[8]
    for k in range(7, -1, -1):
        s = fma(s, z, S[k])
    c = C[9]
    for k in range(8, -1, -1):
        c = fma(c, z, C[k])
    return y * s, c


def box_muller(x):
    """(z_even, z_odd) of one Philox block x = [x0, x1, x2, x3]: the contract of include/pbbi.h, step by step"""
    w1, w2 = (x[1] << 32) | x[0], (x[3] << 32) | x[2]
    u1 = ((w1 >> 12) + 0.5) * 2.0 ** -52
    k2 = w2 >> 11
    n = (k2 + (1 << 50)) >> 51
    y = float(k2 - (n << 51)) * 2.0 ** -51
    sn, cs = sincos_quarter(y)
    r = math.sqrt(-2.0 * log_u(u1))
    c, s = [(cs, sn), (-sn, cs), (-cs, -sn), (sn, -cs)][n & 3]
    return r * c, r * s


if __name__ == "__main__":
    print("// sin((pi/2) y) = y * sum_k S[k] y^(2k),  |y| <= 1/2")
    print("S =", ", ".join(x.hex() for x in S))
    print("// cos((pi/2) y) = sum_k C[k] y^(2k)")
    print("C =", ", ".join(x.hex() for x in C))
    print("Lg =", ", ".join(x.hex() for x in Lg))
    print("ln2_hi", ln2_hi.hex(), repr(ln2_hi), "ln2_lo", ln2_lo.hex(), repr(ln2_lo))
    random.seed(1)
    worst_log = worst_s = worst_c = 0.0
    getcontext().prec = 40
    for _ in range(20000):
        k1 = random.getrandbits(52)
        if random.random() < 0.2:
            k1 >>= random.randrange(0, 52)
        u = (k1 + 0.5) * 2.0 ** -52
        got, ref = log_u(u), math.log(u)
        if ref != 0:
            worst_log = max(worst_log, abs(got - ref) / math.ulp(ref))
        y = random.uniform(-0.5, 0.5)
        s, c = sincos_quarter(y)
        # reference: Taylor series of the exact angle (pi/2) y in 40-digit decimals
        x = Decimal(y) * h
        rs = sum((-1) ** k * x ** (2 * k + 1) / fact(2 * k + 1) for k in range(14))
        rc = sum((-1) ** k * x ** (2 * k) / fact(2 * k) for k in range(14))
        if y:
            worst_s = max(worst_s, abs(Decimal(s) - rs) / Decimal(math.ulp(float(rs))))
        worst_c = max(worst_c, abs(Decimal(c) - rc) / Decimal(math.ulp(float(rc))))
    print("worst error in ulp: log %.2f (vs math.log)  sin %.2f  cos %.2f (vs 40-digit series)" %
          (worst_log, float(worst_s), float(worst_c)))
