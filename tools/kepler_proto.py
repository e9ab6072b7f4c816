"""Prototype (NumPy) of the mixed-precision Kepler solve used by the HIP kernel: fp32 Markley starter, one
fp64 fifth-order correction with cheap reciprocals, sincos by rotation. Measures error vs an 80-bit Newton
solution and vs the all-fp64 Markley of the reference. Development aid; not shipped, not imported by tests."""
import numpy as np

f32 = np.float32
PI = np.pi
K0 = 3 * PI**2 / (PI**2 - 6)
K1N = 8 * PI / (5 * (PI**2 - 6))


def truth(M, e):
    M = M.astype(np.longdouble); e = e.astype(np.longdouble)
    E = np.where(e < 0.8, M, np.sign(M) * np.longdouble(PI))
    E = np.where(M == 0, 0, E)
    for _ in range(100):
        E = E - (E - e * np.sin(E) - M) / (1 - e * np.cos(E))
    return E


def markley64(M, e):
    al = (3 * PI**2 + 8 * (PI**2 - PI * np.abs(M)) / (5 * (1 + e))) / (PI**2 - 6)
    d = 3 * (1 - e) + al * e
    q = 2 * al * d * (1 - e) - M * M
    r = 3 * al * d * (d - 1 + e) * M + M**3
    w = np.cbrt((np.abs(r) + np.sqrt(q**3 + r * r))**2)
    E1 = (2 * r * w / (w * w + w * q + q * q) + M) / d
    f2 = e * np.sin(E1); f3 = e * np.cos(E1)
    f0 = E1 - f2 - M; f1 = 1 - f3
    d3 = -f0 / (f1 - f0 * f2 / (2 * f1))
    d4 = -f0 / (f1 + f2 * d3 / 2 + d3 * d3 * f3 / 6)
    d5 = -f0 / (f1 + d4 * f2 / 2 + d4 * d4 * f3 / 6 - d4**3 * f2 / 24)
    return E1 + d5, d5


def starter32(M, e):
    Mf = M.astype(f32); ef = e.astype(f32); omef = (1 - e).astype(f32)
    k1 = (K1N / (1 + e)).astype(f32)
    al = k1 * (f32(PI) - np.abs(Mf)) + f32(K0)
    d = al * ef + f32(3) * omef
    ad = al * d
    M2 = Mf * Mf
    q = f32(2) * ad * omef - M2
    r = Mf * (f32(3) * ad * (d - omef) + M2)
    q2 = q * q
    s = np.maximum(q2 * q + r * r, f32(0))
    x = np.abs(r) + np.sqrt(s)
    # w = x^(2/3) via exp2/log2 (v_log_f32 / v_exp_f32)
    with np.errstate(divide='ignore'):
        w = np.exp2(np.log2(x) * f32(2.0 / 3.0)).astype(f32)
    den = w * (w + q) + q2
    E1 = (f32(2) * r * w / den + Mf) / d
    return E1.astype(np.float64)


def refine(M, e, E1, rcp_bits=26, nr=(1, 1, 2)):
    def rcp(x, n):
        # emulate a hardware reciprocal with rcp_bits of precision followed by n Newton steps
        r = 1.0 / x
        r = r * (1 + (np.random.default_rng(0).uniform(-1, 1, x.shape)) * 2.0**-rcp_bits)
        for _ in range(n):
            r = r + r * (1 - x * r)
        return r
    s1 = np.sin(E1); c1 = np.cos(E1)
    f2 = e * s1; f3 = e * c1
    f0 = (E1 - M) - f2
    f1 = 1 - f3
    d3 = -f0 * rcp(f1 - f0 * f2 * 0.5 * rcp(f1, nr[0]), nr[0])
    d4 = -f0 * rcp(f1 + d3 * (0.5 * f2 + d3 * f3 / 6), nr[1])
    d5 = -f0 * rcp(f1 + d4 * (0.5 * f2 + d4 * (f3 / 6 - d4 * f2 / 24)), nr[2])
    E = E1 + d5
    dd = d5 * d5
    sd = d5 * (1 + dd * (-1 / 6 + dd / 120))
    cm1 = dd * (-0.5 + dd * (1 / 24 - dd / 720))
    sE = s1 + (s1 * cm1 + c1 * sd)
    cE = c1 + (c1 * cm1 - s1 * sd)
    return E, sE, cE, d5


def refine_device(M, e, E1, rcp_bits=23):
    """The correction as octo_device.h: kepler_solve computes it since round 3: ONE hardware reciprocal (Halley's denominator), δ3, then δ4
    and δ5 as one correction step each on the QUOTIENT, δ' = δ − r·(den·δ + f0), with the same crude r — 2 FMAs per division."""
    s1 = np.sin(E1); c1 = np.cos(E1)
    hf2 = 0.5 * e * s1; q24 = hf2 / 12; sf3 = e / 6 * c1; f1 = 1 - e * c1; f0 = (E1 - M) - e * s1
    den3 = f1 * f1 - f0 * hf2
    r3 = (1.0 / den3) * (1 + np.random.default_rng(0).uniform(-1, 1, den3.shape) * 2.0**-rcp_bits)
    r4 = f1 * r3; d3 = -f0 * r4
    den4 = f1 + d3 * (hf2 + d3 * sf3)
    d4 = d3 - r4 * (den4 * d3 + f0)
    den5 = f1 + d4 * (hf2 + d4 * (sf3 - d4 * q24))
    d5 = d4 - r4 * (den5 * d4 + f0)
    E = E1 + d5; dd = d5 * d5
    sd = d5 * (1 + dd * (-1 / 6)); cm1 = dd * (-0.5 + dd / 24)
    return E, s1 + (c1 * sd + s1 * cm1), c1 + (-s1 * sd + c1 * cm1), d5


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    n = 2_000_000
    e = np.concatenate([rng.uniform(0, 1, n // 2), 1 - 10**rng.uniform(-9, -1, n // 2)])
    M = np.concatenate([rng.uniform(-PI, PI, n // 4), 10**rng.uniform(-17, 0.4, n // 4) * rng.choice([-1, 1], n // 4),
                        (PI - 10**rng.uniform(-16, 0, n // 4)) * rng.choice([-1, 1], n // 4), rng.uniform(-PI, PI, n - 3 * (n // 4))])
    M = np.clip(M, -PI, PI)
    rng.shuffle(M)
    Et = truth(M, e)
    E64, d5_64 = markley64(M, e)
    E1 = starter32(M, e)
    E, sE, cE, d5 = refine_device(M, e, E1)      # refine(M, e, E1): the reciprocal-refinement form of rounds 1-2
    cond = 1 - e * np.cos(Et.astype(np.float64))
    err64 = np.abs((E64 - Et).astype(np.float64)) * cond
    err = np.abs((E - Et).astype(np.float64)) * cond
    print("non-finite:", np.sum(~np.isfinite(E)), "max|d5| fp64:", np.nanmax(np.abs(d5_64)), "mixed:", np.nanmax(np.abs(d5)))
    print("residual-weighted error  reference fp64: max %.3e  p99.9 %.3e | mixed: max %.3e p99.9 %.3e" % (
        err64.max(), np.quantile(err64, 0.999), np.nanmax(err), np.nanquantile(err, 0.999)))
    es = np.abs(sE - np.sin(Et).astype(np.float64)); ec = np.abs(cE - np.cos(Et).astype(np.float64))
    es0 = np.abs(np.sin(E64) - np.sin(Et).astype(np.float64))
    print("sinE abs err mixed max %.3e (weighted %.3e) | reference %.3e" % (es.max(), (es * cond).max(), es0.max()))
    worst = np.nanargmax(err)
    print("worst case e=%.17g M=%.17g err=%.3e ref=%.3e d5=%.3e" % (e[worst], M[worst], err[worst], err64[worst], d5[worst]))
    for lo, hi in [(0, .5), (.5, .9), (.9, .99), (.99, .999), (.999, .999999), (.999999, 1)]:
        m = (e >= lo) & (e < hi)
        print(f"e in [{lo},{hi}): ref max {err64[m].max():.2e}  mixed max {np.nanmax(err[m]):.2e}  max|d5| {np.nanmax(np.abs(d5[m])):.2e}  unweighted dE ref {np.abs((E64-Et)[m]).max():.2e} mixed {np.nanmax(np.abs((E-Et)[m])):.2e}")
