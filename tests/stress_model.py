"""Randomised sweep of the whole callback (octo_model_logpost: invlink, priors with Jacobian, UniformCircular, θ_at_epoch_to_tperi
in both bases, likelihood, ∇θ_t) against the oracle's restatement, over random systems from stress_parity.random_system with
random standard-parameterisation models on top. Run on a GPU box:  python tests/stress_model.py [n] [seed]."""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob, gpu_binding as gb
from stress_parity import random_system, draw_system
capi = gb.capi

P_ = lambda kind, p0=0.0, p1=0.0, lo=None, hi=None: dict(kind=kind, p0=p0, p1=p1, lo=lo, hi=hi)
S_ = lambda kind, i0=0, i1=0, flags=0, value=0.0: dict(kind=kind, i0=i0, i1=i1, flags=flags, value=value)


def random_model(rng, obs, planets):
    priors, esrc, nsrc = [], [], []
    def new(pr):
        priors.append(pr); return len(priors) - 1
    iM = new(P_(3, 1.2, 0.1, 0.3, None)); iplx = new(P_(2, 45.0, 0.5))
    for ip, pl in enumerate(planets):
        ti = pl["orbit_kind"] == 2
        a_lo = 2.0 + 6 * ip
        row = [None] * 9
        if ti:
            for k in (0, 2, 3, 4): row[k] = S_(1, new(P_(2, 0.0, 300.0)))
        else:
            row[0] = S_(1, new(P_(1, a_lo, a_lo + 4.0) if rng.random() < 0.5 else P_(0, a_lo, a_lo + 4.0)))
            row[2] = S_(1, new(P_(4)))
            for k in (3, 4):
                i0, i1 = new(P_(2, 0.0, 1.0)), new(P_(2, 0.0, 1.0))
                row[k] = S_(2, i0, i1, 1, 2 * np.pi)
        row[1] = S_(1, new(P_(0, 0.0, 0.8)))
        if rng.random() < 0.6:
            i0, i1 = new(P_(2, 0.0, 1.0)), new(P_(2, 0.0, 1.0))
            row[5] = S_(3, i0, i1, 1 | (2 if ti else 0), 50000.0)
        else:
            row[5] = S_(1, new(P_(0, 47000.0, 53000.0)))
        row[6] = S_(1, iM); row[7] = S_(1, iplx)
        row[8] = S_(1, new(P_(1, 1.0, 40.0)))
        esrc += row
    for o in obs:
        k = o["kind"]
        if k in (0, 1, 5, 6):
            nsrc += [S_(1, new(P_(1, 0.1, 5.0))) if rng.random() < 0.6 else S_(0), S_(1, new(P_(2, 1.0, 0.01))) if rng.random() < 0.4 else S_(0, value=1.0),
                     S_(1, new(P_(2, 0.0, 0.02))) if rng.random() < 0.4 else S_(0)]
        elif k == 7:
            nsrc += [S_(1, new(P_(2, 4.3, 0.5))), S_(1, new(P_(2, -2.0, 0.5))), S_(0)]
        else:
            # third row: the trend coefficient (OCTO_NU_RV_TREND) where the table carries a basis column
            nsrc += [S_(1, new(P_(2, 0.0, 20.0))) if k != 3 else S_(0), S_(1, new(P_(1, 0.1, 10.0))),
                     S_(1, new(P_(2, 0.0, 2.0))) if o.get("extra") is not None else S_(0)]
    return priors, esrc, nsrc


def check_model(rng, lib, P=None, W=None, evaluate=True):
    """One random system + random standard-parameterisation model through octo_model_logpost vs the oracle. Returns None if the
    draw has more than 64 parameters, else (ok, e_lp, e_grad, loose, description)."""
    obs, planets, elems, _ = draw_system(rng, invalid=False, P=P, W=W)
    priors, esrc, nsrc = random_model(rng, obs, planets)
    D = len(priors)
    if D > 64:
        return None
    W = elems.shape[1]
    th = rng.normal(0, 1, (D, W))
    th[1] = 45.0 + 0.5 * rng.normal(0, 1, W)
    for d, pr in enumerate(priors):      # identity-link priors: draw in their natural scale
        if pr["kind"] == 2: th[d] = pr["p0"] + pr["p1"] * rng.normal(0, 1, W)
    desc = f"P={len(planets)} bases={[p['orbit_kind'] for p in planets]} kinds={[o['kind'] for o in obs]} D={D} W={W}"
    if not evaluate:      # the draw has consumed its random numbers: a caller replaying a sweep skips to the cases it wants
        return None
    pr_c, es_c, ns_c = ob.make_priors(priors), ob.make_sources(esrc), ob.make_sources(nsrc)
    path = gb.GpuPath(obs, planets)
    m = C.c_void_p()
    st = lib.octo_model_create(path.ctx, path.ds, pr_c, D, es_c, ns_c, C.byref(m))
    assert st == 0, (st, lib.octo_last_error(path.ctx))
    thc = np.ascontiguousarray(th); lp = np.empty(W); g = np.empty_like(thc); lp0 = np.empty(W)
    gb.poison(path.ctx)
    assert lib.octo_model_logpost(path.ctx, m, capi._dptr(thc), W, W, capi._dptr(lp), capi._dptr(g)) == 0
    gb.poison(path.ctx)
    assert lib.octo_model_logpost(path.ctx, m, capi._dptr(thc), W, W, capi._dptr(lp0), None) == 0
    lib.octo_model_destroy(m); path.close()
    lp_o, g_o = ob.oracle_model_logpost(obs, planets, pr_c, es_c, ns_c, thc, n_threads=0)
    ok = np.isfinite(lp_o) & (lp_o > -1e300)
    same = np.array_equal(lp, lp0) and np.array_equal(np.isfinite(lp) & (lp > -1e300), ok)
    e_lp = np.max(np.abs(lp[ok] - lp_o[ok]) / np.maximum(1, np.abs(lp_o[ok]))) if ok.any() else 0.0
    scale = np.maximum(np.abs(g_o[:, ok]).max(axis=1, keepdims=True), 1e-10 * np.abs(g_o[:, ok]).max()) if ok.any() else 1.0
    e_g = np.max(np.abs(g[:, ok] - g_o[:, ok]) / scale) if ok.any() else 0.0
    loose = any(o["kind"] == 3 for o in obs) or any(p["orbit_kind"] == 2 for p in planets)
    lim = (1e-9, 1e-7) if loose else (1e-12, 1e-9)
    return bool(same and e_lp < lim[0] and e_g < lim[1]), float(e_lp), float(e_g), loose, desc


def main():
    n_sys = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    lib = capi.load_library()
    bad = 0; worst = [0.0, 0.0]
    for k in range(n_sys):
        r = check_model(rng, lib)
        if r is None: continue
        good, e_lp, e_g, loose, desc = r
        bad += not good
        if not loose: worst = [max(worst[0], e_lp), max(worst[1], e_g)]
        print(f"{k:3d} {desc}: lp {e_lp:.1e} grad/scale {e_g:.1e}{'' if good else '   <-- FAIL'}", flush=True)
    print(f"worst (no marginalised RV, no Thiele-Innes): lp {worst[0]:.2e} grad {worst[1]:.2e}; failures {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
