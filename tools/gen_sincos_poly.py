"""Generate near-minimax polynomials (Chebyshev interpolation at 60 digits, then rounded to double) for
sin(h)/h and cos(h) as polynomials in u = h^2 on |h| <= HMAX, used by the half-angle sincos of the HIP
kernel (octo_device.h). Prints C arrays and the achieved error when evaluated in double. Development aid."""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
HMAX = mp.mpf("1.59")          # |E1|/2 <= (π + starter overshoot)/2 ≈ 1.5715; margin


def cheb_fit(f, a, b, n):
    # interpolate f on [a,b] at n Chebyshev nodes -> monomial coefficients in u
    nodes = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, u in enumerate(nodes):
        for j in range(n):
            A[i, j] = u ** j
        y[i] = f(u)
    c = mp.lu_solve(A, y)
    return [c[j] for j in range(n)]


def fs(u):  # sin(sqrt(u))/sqrt(u)
    if u == 0:
        return mp.mpf(1)
    h = mp.sqrt(u)
    return mp.sin(h) / h


def fc(u):
    return mp.cos(mp.sqrt(u))


def horner(c, u):
    r = np.full_like(u, c[-1])
    for k in range(len(c) - 2, -1, -1):
        r = r * u + c[k]
    return r


if __name__ == "__main__":
    umax = HMAX ** 2
    for ns, nc in [(9, 10), (10, 10), (10, 11), (11, 11)]:
        cs = cheb_fit(fs, mp.mpf(0), umax, ns)
        cc = cheb_fit(fc, mp.mpf(0), umax, nc)
        # approximation error at 60 digits with double-rounded coefficients
        csd = [float(x) for x in cs]
        ccd = [float(x) for x in cc]
        worst_s = worst_c = mp.mpf(0)
        for k in range(2001):
            u = umax * k / 2000
            ps = sum(mp.mpf(csd[j]) * u ** j for j in range(ns))
            pc = sum(mp.mpf(ccd[j]) * u ** j for j in range(nc))
            worst_s = max(worst_s, abs(ps - fs(u)))
            worst_c = max(worst_c, abs(pc - fc(u)))
        h = np.linspace(-float(HMAX), float(HMAX), 400001)
        u = h * h
        s = h * horner(csd, u)
        c = horner(ccd, u)
        es = np.max(np.abs(s - np.sin(h))); ec = np.max(np.abs(c - np.cos(h)))
        sx = 2 * s * c; cx = 1 - 2 * s * s
        e2s = np.max(np.abs(sx - np.sin(2 * h))); e2c = np.max(np.abs(cx - np.cos(2 * h)))
        print(f"ns={ns} nc={nc}: approx err sin {float(worst_s):.2e} cos {float(worst_c):.2e} | double eval err sin(h) {es:.2e} cos(h) {ec:.2e} | sin(2h) {e2s:.2e} cos(2h) {e2c:.2e}")
        if (ns, nc) == (10, 11):
            print("SIN:", ", ".join(f"{x!r}" for x in csd))
            print("COS:", ", ".join(f"{x!r}" for x in ccd))
