"""Batches of the reference's own GP at the reference's own size -- the retro loop
(north/retrospective_forecasts/September1st_retro.py:176-248: 3 regions x ~40 years, n = year - 1979 <= 45) times the
20 x 20 (l, sn~) grid of north/June1st.py:210-211 -- through ``sigp_small_upload`` / ``sigp_small_run``: one workgroup per
fit, one launch for the whole list (include/sigp.h).

Host side: data sets are staged in factored form so that the device never needs ``expm``:

  expm="eigh"  M = Q diag(lam) Q^T once per data set;  A = [X ; Xs] Q, weights exp(l lam) formed on the device
               for every l of a grid.
  expm="pade"  Sigma~ = scipy.linalg.expm(l M) per (data set, l) as the reference does (north/June1st.py:264), then
               Sigma~ = U diag(s) U^T;  A = [X ; Xs] U, weights s.  Reproduces the reference's numbers also for its
               extreme table entry l = 3.1e10, at one host eigendecomposition per (data set, l).

``run(grad=True)`` is the MLII closure (north/June1st.py:235-257) for the whole list in the same single launch: the value, the
reference's own "gradient" formulae (:248-252) and the exact derivative (``sigp_small_run_grad``).  M Sigma~ needs no second
matrix: in the eigenbasis it is the reweighting lam_k exp(l lam_k); "pade" sets carry u_k^T M u_k * s_k instead.
"""
import numpy as np

from . import _lib as L
from .features import laplacian_M, sigma_tilde

NMAX, MMAX = 128, 8


class SmallBatch:
    def __init__(self, gp):
        if gp.kernel != "netdiffusion" or gp.dtype != "f64":
            raise ValueError("SmallBatch runs the reference kernel on the fp64 engine")
        self.gp = gp
        self._data = []          # (X, y, Xs, M)
        self._eig = {}           # ds -> (lam, Q)
        self._sets = []          # device sets: (A, y, lam, mode, dlam)
        self._set_of = {}        # (ds, None) or (ds, ell) -> device set index
        self._fits = []          # (device set, ell, sn)
        self._packed = None      # the fit list as arrays (rebuilt after add_fit)
        self._uploaded = 0

    # ---- building the list ---------------------------------------------------------------------------------------
    def add_dataset(self, X, y, Xs=None, M=None):
        """One (region, year): X [n, N], y [n] or [n, 1], Xs [m, N] (m <= 8) or None, M = the graph Laplacian
        (north/June1st.py:231-233; computed from X when omitted).  Returns the data-set id."""
        X = L.f64(X, 2)
        y = L.f64(np.asarray(y).reshape(-1), 1)
        n, N = X.shape
        if y.shape[0] != n:
            raise ValueError("X has %d rows but y has %d" % (n, y.shape[0]))
        if not (1 <= n <= NMAX):
            raise ValueError("this path takes 1 <= n <= %d training rows (got %d); larger fits go through GPR.fit" % (NMAX, n))
        if Xs is None:
            Xs = np.zeros((0, N))
        Xs = L.f64(np.atleast_2d(Xs), 2)
        if Xs.shape[1] != N or Xs.shape[0] > MMAX:
            raise ValueError("Xs must be [m <= %d, %d]" % (MMAX, N))
        M = laplacian_M(X) if M is None else L.f64(M, 2)
        if M.shape != (N, N):
            raise ValueError("M must be %dx%d" % (N, N))
        self._data.append((X, y, Xs, M))
        return len(self._data) - 1

    def _device_set(self, ds, ell, expm):
        key = (ds, None) if expm == "eigh" else (ds, float(ell))
        idx = self._set_of.get(key)
        if idx is not None:
            return idx
        X, y, Xs, M = self._data[ds]
        XX = np.vstack([X, Xs])
        if expm == "eigh":
            if ds not in self._eig:
                lam, Q = np.linalg.eigh(0.5 * (M + M.T))
                self._eig[ds] = (np.minimum(lam, 0.0), Q)      # the exact spectrum is <= 0
            lam, Q = self._eig[ds]
            dev = (np.ascontiguousarray(XX @ Q), y, np.ascontiguousarray(lam), 0, np.zeros_like(lam))
        else:
            Sig = sigma_tilde(M, float(ell))
            if not np.all(np.isfinite(Sig)):
                raise FloatingPointError("expm(l M) overflowed")
            s, U = np.linalg.eigh(0.5 * (Sig + Sig.T))
            # M and Sigma~ = expm(l M) share eigenvectors: M Sigma~ = U diag(mu s) U^T with mu_k = u_k^T M u_k (the MLII gradient's
            # d Sigma/dl, north/June1st.py:248); inside a degenerate cluster of s the choice of U does not matter where s ~ 0
            mu = np.einsum("ik,ij,jk->k", U, M, U)
            dev = (np.ascontiguousarray(XX @ U), y, np.ascontiguousarray(s), 1, np.ascontiguousarray(mu * np.maximum(s, 0.0)))
        self._sets.append(dev)
        self._set_of[key] = len(self._sets) - 1
        return self._set_of[key]

    def add_fit(self, ds, ell, sn_tilde, expm="eigh"):
        """Queue one fit of data set ``ds`` at (l, sn~).  Returns its index in the result arrays."""
        if expm not in ("eigh", "pade"):
            raise ValueError("expm must be 'eigh' or 'pade'")
        if not (float(ell) > 0) or not (float(sn_tilde) >= 0):
            raise ValueError("ell > 0 and sn_tilde >= 0 required")
        self._fits.append((self._device_set(ds, ell, expm), float(ell), float(sn_tilde)))
        self._packed = None
        return len(self._fits) - 1

    # ---- device ---------------------------------------------------------------------------------------------------
    def upload(self):
        """Stage every data set queued so far in HBM (flat pools + offsets)."""
        sets = self._sets
        if not sets:
            raise RuntimeError("no fits queued")
        n = np.array([s[1].shape[0] for s in sets], dtype=np.int64)
        N = np.array([s[0].shape[1] for s in sets], dtype=np.int64)
        m = np.array([s[0].shape[0] - s[1].shape[0] for s in sets], dtype=np.int64)
        mode = np.array([s[3] for s in sets], dtype=np.int32)
        a_off = np.concatenate([[0], np.cumsum((n + m) * N)[:-1]]).astype(np.int64)
        y_off = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64)
        l_off = np.concatenate([[0], np.cumsum(N)[:-1]]).astype(np.int64)
        A = np.concatenate([s[0].reshape(-1) for s in sets])
        y = np.concatenate([s[1] for s in sets])
        lam = np.concatenate([s[2] for s in sets])
        gp = self.gp
        gp._check(gp._lib.sigp_small_upload(gp._h, len(sets), L.iptr(n), L.iptr(N), L.iptr(m), L.iptr(mode), L.ptr(A), L.iptr(a_off),
                                            L.ptr(y), L.iptr(y_off), L.ptr(lam), L.iptr(l_off)), "small_upload")
        if mode.any():
            dlam = np.concatenate([s[4] for s in sets])
            gp._check(gp._lib.sigp_small_set_dweights(gp._h, L.ptr(dlam), len(dlam)), "small_set_dweights")
        self._uploaded = len(sets)
        self._mmax = int(m.max())

    def clear_fits(self):
        """Forget the queued fits; the data sets (and what is staged in HBM) stay."""
        self._fits = []
        self._packed = None

    def run(self, grad=False):
        """All queued fits in one launch -> dict(sigma_f, nlml, info, sigma_n, mean [F, mmax], var [F, mmax]); with ``grad=True``
        also grad_ref [F, 2] (the reference's MLII formulae, north/June1st.py:248-252) and grad_exact [F, 2] (the derivative of the
        profiled nlML w.r.t. (log l, log sn~)); +inf where K~ is not positive definite (:254-256)."""
        if self._uploaded != len(self._sets):
            self.upload()
        F = len(self._fits)
        if self._packed is None:
            arr = np.asarray(self._fits, dtype=np.float64)
            self._packed = (np.ascontiguousarray(arr[:, 0], dtype=np.int64), np.ascontiguousarray(arr[:, 1]), np.ascontiguousarray(arr[:, 2]))
        si, ell, sn = self._packed
        ms = max(self._mmax, 1)
        out = np.zeros((F, 8 if grad else 4)); mean = np.full((F, ms), np.nan); var = np.full((F, ms), np.nan)
        gp = self.gp
        fn = gp._lib.sigp_small_run_grad if grad else gp._lib.sigp_small_run
        gp._check(fn(gp._h, F, L.iptr(si), L.ptr(ell), L.ptr(sn), L.ptr(out), L.ptr(mean), L.ptr(var), ms), "small_run")
        gp._fitted = False
        res = dict(sigma_f=out[:, 0], nlml=out[:, 1], info=out[:, 2].astype(np.int64), sigma_n=out[:, 3],
                   mean=mean[:, :self._mmax], var=var[:, :self._mmax])
        if grad:
            res["grad_ref"], res["grad_exact"] = out[:, 4:6].copy(), out[:, 6:8].copy()
        return res
