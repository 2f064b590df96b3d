"""Regenerate csrc/pdt_sincostab.h (the 110 x 4 double-double table of sin / cos at k/128 that glibc's double sin/cos use)
from the definition, with 90-digit decimal arithmetic.  usage: python tools/gen_sincostab.py > table.txt"""
from decimal import Decimal, getcontext

getcontext().prec = 90


def sin_cos(x):
    s = c = Decimal(0)
    t, k = Decimal(1), 0
    while abs(t) > Decimal(10) ** -85:
        if k % 2 == 0:
            c += t if (k // 2) % 2 == 0 else -t
        else:
            s += t if (k // 2) % 2 == 0 else -t
        k += 1
        t = t * x / k
    return s, c


for k in range(110):
    s, c = sin_cos(Decimal(k) / Decimal(128))
    sn = float(s); ssn = float(s - Decimal(sn)); cs = float(c); ccs = float(c - Decimal(cs))     # float(Decimal) rounds to nearest
    print("    " + ", ".join(v.hex() for v in (sn, ssn, cs, ccs)) + ",")
