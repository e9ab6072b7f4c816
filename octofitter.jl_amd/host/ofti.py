"""
Host-side mirror of `ofti_linear_solve` (src/parameterizations.jl:318-405), batched over walkers, on the HIP path.

    solver = OftiLinearSolver(epochs, ra_data, dec_data, σ_ra, σ_dec, cor, σ_ABFG)
    res = solver(e, a, tp, M, plx)        # arrays of length W -> dict(A, B, F, G, log_marginal_likelihood)
    ofti_linear_solve(epochs, ra_data, dec_data, σ_ra, σ_dec, cor, σ_ABFG, e, a, tp, M, plx)   # one-shot, reference signature

This is the likelihood the reference's OFTI example feeds to `octofit_rejection` with 1e6 prior draws
(examples/ofti_rejection_sampling.jl:78-85,108) — a natural walker batch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class OftiLinearSolver:
    def __init__(self, epochs, ra_data, dec_data, σ_ra, σ_dec, cor, σ_ABFG, device: int = 0, consts=None):
        self.lib = capi.load_library()
        cols = [np.ascontiguousarray(v, dtype=np.float64) for v in (epochs, ra_data, dec_data, σ_ra, σ_dec)]
        n = len(cols[0])
        if any(len(c) != n for c in cols):
            raise ValueError("The columns in the input data do not all have the same length")
        cc = None if cor is None else np.ascontiguousarray(cor, dtype=np.float64)
        self._ctx = C.c_void_p()
        st = self.lib.octo_ctx_create(C.byref(self._ctx), int(device))
        if st != capi.OCTO_OK:
            raise capi.OctoError(st, "octo_ctx_create")
        if consts is not None:
            self._check(self.lib.octo_consts_set(self._ctx, C.byref(consts)), "octo_consts_set")
        self._h = C.c_void_p()
        self._check(self.lib.octo_ofti_create(self._ctx, *[capi._dptr(c) for c in cols], capi._dptr(cc), n, float(σ_ABFG),
                                              C.byref(self._h)), "octo_ofti_create")

    def _check(self, status, what):
        if status != capi.OCTO_OK:
            raise capi.OctoError(status, f"{what}: {(self.lib.octo_last_error(self._ctx) or b'').decode()}")

    def __call__(self, e, a, tp, M, plx):
        nl = np.ascontiguousarray(np.stack(np.broadcast_arrays(*[np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (e, a, tp, M, plx)])))
        W = nl.shape[1]
        abfg = np.empty((4, W))
        lm = np.empty(W)
        self._check(self.lib.octo_ofti_eval(self._ctx, self._h, capi._dptr(nl), W, W, capi._dptr(abfg), capi._dptr(lm)), "octo_ofti_eval")
        return dict(A=abfg[0], B=abfg[1], F=abfg[2], G=abfg[3], log_marginal_likelihood=lm)

    def eval_device(self, nl_t, stream=None):
        """nl_t: torch float64 CUDA tensor [5, W]; returns (abfg [4, W], logml [W]) tensors, asynchronous."""
        import torch
        W = nl_t.shape[1]
        abfg = torch.empty((4, W), dtype=torch.float64, device=nl_t.device)
        lm = torch.empty(W, dtype=torch.float64, device=nl_t.device)
        if stream is None:
            stream = torch.cuda.current_stream(nl_t.device).cuda_stream
        self._check(self.lib.octo_ofti_eval_device(self._ctx, self._h, nl_t.data_ptr(), W, W, abfg.data_ptr(), lm.data_ptr(),
                                                   C.c_void_p(stream)), "octo_ofti_eval_device")
        return abfg, lm

    def close(self):
        if getattr(self, "_h", None):
            self.lib.octo_ofti_destroy(self._h); self._h = None
        if getattr(self, "_ctx", None):
            self.lib.octo_ctx_destroy(self._ctx); self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ofti_linear_solve(epochs, ra_data, dec_data, σ_ra, σ_dec, cor, σ_ABFG, e, a, tp, M, plx, device: int = 0):
    solver = OftiLinearSolver(epochs, ra_data, dec_data, σ_ra, σ_dec, cor, σ_ABFG, device=device)
    try:
        return solver(e, a, tp, M, plx)
    finally:
        solver.close()
