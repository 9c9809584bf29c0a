#!/usr/bin/env python3
"""Coefficients of the degree-2 minimax polynomial of 2^(r/N) on r in [0, 1] used by the Sinkhorn sweep
(SK2_C0..C2 in csrc/sinkhorn.hip).  Remez exchange in 60-digit arithmetic (mpmath); prints hex doubles.
    python tools/exp2_minimax.py [N=4096] [degree=2]"""
import sys

import mpmath as mp

mp.mp.dps = 60
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def f(r):
    return mp.power(2, r / N)


n = deg + 2
xs = [(1 - mp.cos(mp.pi * i / (n - 1))) / 2 for i in range(n)]
grid = [mp.mpf(i) / 4000 for i in range(4001)]
for _ in range(30):
    A, b = mp.matrix(n, n), mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(deg + 1):
            A[i, j] = x ** j
        A[i, deg + 1] = (-1) ** i
        b[i] = f(x)
    sol = mp.lu_solve(A, b)
    c = [sol[j] for j in range(deg + 1)]
    vals = [sum(c[j] * x ** j for j in range(deg + 1)) - f(x) for x in grid]
    ext = [0] + [i for i in range(1, 4000) if (vals[i] - vals[i - 1]) * (vals[i + 1] - vals[i]) < 0] + [4000]
    if len(ext) != n:
        break
    xs = [grid[i] for i in ext]
print(f"N = {N}, degree {deg}: max |error| = {mp.nstr(max(abs(v) for v in vals), 4)}")
for j, cj in enumerate(c):
    print(f"  c{j} = {float(cj).hex()}   ({mp.nstr(cj, 22)})")
