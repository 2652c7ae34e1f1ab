"""``GPR``: the fit / predict / nlml call sites that replace the reference's inline GP block
(north/June1st.py:231-277 and its 13 byte-identical siblings).

A script's block becomes::

    gp = GPR(kernel="netdiffusion")
    gp.fit(X, y, l_init[k], sigma_init[k], M=M, Xs=Xs)      # :264-271
    fmean, fvar = gp.predict(Xs)                            # :272-277  (fvar includes sigma_n)

All numerics run in HIP kernels behind libsigp.so (include/sigp.h).  There is no CPU fallback: without
the library or a GPU the constructor raises.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .features import SigmaEigh, laplacian_M, sigma_tilde


class LinAlgError(np.linalg.LinAlgError):
    """Raised where the reference's live block raises ``np.linalg.LinAlgError`` (non-SPD K~ at
    north/June1st.py:265); carries the LAPACK-style pivot index in ``.info``."""

    def __init__(self, msg, info=0):
        super().__init__(msg)
        self.info = info


class GPR:
    def __init__(self, kernel="netdiffusion", dtype="f64", device=0, outer_blocks=None, lookahead=None, schedule=None,
                 panel_mode=None, expm="pade"):
        if kernel not in L.KERNEL_IDS:
            raise ValueError("kernel must be one of %s" % sorted(L.KERNEL_IDS))
        if dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' or 'f32'")
        if dtype == "f32" and kernel == "netdiffusion":
            raise ValueError("the fp32 engine (fp32 factor + fp64 iterative refinement) covers the RBF / Matern kernels only")
        if expm not in ("pade", "eigh"):
            raise ValueError("expm must be 'pade' (scipy.linalg.expm per call, the reference's numbers) or 'eigh' (one eigendecomposition per data set)")
        self._expm = expm              # reference kernel only: how Sigma~ = expm(l M) is formed
        self._eig = None
        self.dtype = dtype
        self.kernel = kernel
        self._kid = L.KERNEL_IDS[kernel]
        self._lib = L.load()
        h = C.c_void_p()
        rc = self._lib.sigp_create(C.byref(h), int(device), 0 if dtype == "f64" else 1)
        if rc != L.OK:
            raise L.SigpError("sigp_create(device=%d) failed (rc=%d): no usable MI355X / HIP runtime [%s]; there is no CPU fallback"
                              % (device, rc, L.runtime_info()))
        self._h = h
        self.device = device
        self._has_data = False
        self._fitted = False
        self._ride = None
        if outer_blocks is not None:
            self.set_option("outer_blocks", outer_blocks)
        if lookahead is not None:
            self.set_option("lookahead", int(bool(lookahead)))
        if schedule is not None:      # "right" | "left": outer schedule of the blocked Cholesky (same factor, bit for bit)
            self.set_option("schedule", {"right": 0, "left": 1}[schedule])
        if panel_mode is not None:    # "recursive" | "strips" | "auto": how the rows below a panel's top block are solved
            self.set_option("panel_mode", {"recursive": 0, "strips": 1, "auto": 2}[panel_mode])

    # ---- plumbing ------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc == L.OK:
            return
        msg = self._lib.sigp_last_error(self._h)
        msg = msg.decode() if msg else ""
        if rc == L.NOT_SPD:
            raise LinAlgError("%s: %s" % (what, msg or "Matrix is not positive definite"))
        if rc == L.BAD_ARG:
            raise ValueError("%s: %s" % (what, msg))
        raise L.SigpError("%s: %s" % (what, msg))

    def set_option(self, name, value):
        self._check(self._lib.sigp_set_option(self._h, name.encode(), int(value)), "set_option")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sigp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- data ----------------------------------------------------------------------------------
    def set_data(self, X, y, M=None, Xs=None):
        """Stage the enclosing-scope variables of the reference block (X, y, M, Xs) in HBM."""
        X = L.f64(X, 2)
        y = L.f64(np.asarray(y).reshape(-1), 1)
        if y.shape[0] != X.shape[0]:
            raise ValueError("X has %d rows but y has %d" % (X.shape[0], y.shape[0]))
        self._check(self._lib.sigp_set_train(self._h, L.ptr(X), X.shape[0], X.shape[1], X.shape[1], L.ptr(y)), "set_train")
        self.n, self.d = X.shape
        self._X, self._y = X, y
        self._M = None
        if self.kernel == "netdiffusion":
            self._M = laplacian_M(X) if M is None else L.f64(M, 2)
            if self._M.shape != (self.d, self.d):
                raise ValueError("M must be %dx%d" % (self.d, self.d))
            self._eig = SigmaEigh(self._M) if self._expm == "eigh" else None     # one eigh per data set
        self._set_ride(Xs)
        self._has_data = True
        self._fitted = False

    def _sigma(self, ell, with_derivative=False):
        """Sigma~ = expm(l M) (north/June1st.py:264) and, for the MLII gradient, M Sigma~ (:243-244 d Sigma/dl)."""
        if self._eig is not None:
            Sig = self._eig.sigma(ell)
            return (Sig, self._eig.msigma(ell)) if with_derivative else Sig
        Sig = sigma_tilde(self._M, ell)
        return (Sig, self._M @ Sig) if with_derivative else Sig

    def _set_ride(self, Xs):
        if Xs is None:
            self._ride = None
            self._check(self._lib.sigp_set_test(self._h, None, 0, 0), "set_test")
            return
        Xs = L.f64(np.atleast_2d(Xs), 2)
        if Xs.shape[1] != self.d:
            raise ValueError("Xs must have %d columns" % self.d)
        if Xs.shape[0] > (L.MAX_RIDE if self.dtype == "f64" else 3):
            self._ride = None      # too many to ride along: predict() takes the general path
            self._check(self._lib.sigp_set_test(self._h, None, 0, 0), "set_test")
            return
        self._check(self._lib.sigp_set_test(self._h, L.ptr(Xs), Xs.shape[0], Xs.shape[1]), "set_test")
        self._ride = Xs

    # ---- fit (north/June1st.py:264-271) --------------------------------------------------------
    def fit(self, X, y, ell, sn_tilde, M=None, Xs=None):
        """Kernel build -> Cholesky -> A~ -> profiled sigma_f.  ``Xs`` (optional, <= 127 rows) rides along
        the factorisation so the following ``predict(Xs)`` costs nothing extra.  Raises LinAlgError
        if K~ is not positive definite, as the reference's live block does."""
        self.set_data(X, y, M=M, Xs=Xs)
        return self.refit(ell, sn_tilde)

    def refit(self, ell, sn_tilde):
        """Fit again on the staged data with new hyper-parameters (grid search / optimiser loop)."""
        if not self._has_data:
            raise RuntimeError("refit: no data staged; call fit() or set_data() first")
        out = np.zeros(4)
        m = 0 if self._ride is None else self._ride.shape[0]
        mean = np.zeros(max(m, 1))
        var = np.zeros(max(m, 1))
        Sig = None
        if self.kernel == "netdiffusion":
            Sig = L.f64(self._sigma(float(ell)), 2)
            self._Sigma_tilde = Sig
        self._fitted = False
        rc = self._lib.sigp_fit_predict(self._h, self._kid, float(ell), float(sn_tilde), L.ptr(Sig),
                                        0 if Sig is None else Sig.shape[1], L.ptr(out), L.ptr(mean), L.ptr(var))
        self.info_ = int(out[2]) if rc in (L.OK, L.NOT_SPD) else -1
        if rc == L.NOT_SPD:
            raise LinAlgError("Matrix is not positive definite (pivot %d)" % self.info_, self.info_)
        self._check(rc, "fit")
        self.ell_, self.sn_tilde_ = float(ell), float(sn_tilde)
        self.sigma_f_, self.nlml_, self.sigma_n_ = float(out[0]), float(out[1]), float(out[3])
        self._ride_mean, self._ride_var = mean[:m].copy(), var[:m].copy()
        self._fitted = True
        return self

    # ---- predict (north/June1st.py:272-277) ----------------------------------------------------
    def predict(self, Xs):
        """(fmean [m], fvar [m]); fvar is the variance of y*, i.e. includes sigma_n (:273, :277)."""
        if not self._fitted:
            raise RuntimeError("predict: call fit() first")
        Xs = L.f64(np.atleast_2d(Xs), 2)
        if Xs.shape[1] != self.d:
            raise ValueError("Xs must have %d columns" % self.d)
        if self._ride is not None and Xs.shape == self._ride.shape and np.array_equal(Xs, self._ride):
            return self._ride_mean.copy(), self._ride_var.copy()
        m = Xs.shape[0]
        mean, var = np.zeros(m), np.zeros(m)
        self._check(self._lib.sigp_predict(self._h, L.ptr(Xs), m, Xs.shape[1], L.ptr(mean), L.ptr(var)), "predict")
        return mean, var

    # ---- MLII (north/June1st.py:235-257) -------------------------------------------------------
    def nlml(self, theta, grad="ref"):
        """``MLII(hyperparameters)``: theta = (log l, log sn~) -> (nlML, grad[2]).

        grad='ref'   the reference's own formulae (:248-252; NOT the derivative of nlML, SURVEY App. C-7;
                     defined for the reference kernel only),
        grad='exact' the analytic derivative of the profiled nlML (what an optimiser should be given),
        grad=None    value only (second entry None).
        A non-SPD K~ or an overflowing exp(theta) gives ``(inf, [inf, inf])`` like the reference's except
        branch (:254-256).  Everything O(n^3) runs on the device (K13/K14: tr(K~^-1 dK~) from a lockstep
        forward block solve of the identity + one SYRK)."""
        if not self._has_data:
            raise RuntimeError("nlml: no data staged; call fit() or set_data() first")
        if grad not in (None, "ref", "exact"):
            raise ValueError("grad must be None, 'ref' or 'exact'")
        if grad == "ref" and self.kernel != "netdiffusion":
            raise ValueError("grad='ref' is defined for the reference kernel only")
        theta = L.f64(np.asarray(theta, dtype=np.float64).reshape(2), 1)
        inf2 = (np.inf, np.asarray([np.inf, np.inf]))
        with np.errstate(over="ignore"):
            ell = float(np.exp(theta[0]))
        if not np.isfinite(ell) or not np.isfinite(np.exp(theta[1])):
            return inf2
        Sig = MSig = None
        if self.kernel == "netdiffusion":
            try:
                with np.errstate(over="raise", invalid="raise"):
                    Sig, MSig = self._sigma(ell, with_derivative=True)
                    Sig, MSig = L.f64(Sig, 2), L.f64(MSig, 2)
            except (ValueError, OverflowError, FloatingPointError):
                return inf2
            if not (np.all(np.isfinite(Sig)) and np.all(np.isfinite(MSig))):
                return inf2
        mode = {None: 0, "ref": 1, "exact": 2}[grad]
        val = C.c_double()
        g = np.zeros(2)
        self._fitted = False
        rc = self._lib.sigp_nlml_grad(self._h, self._kid, L.ptr(theta), L.ptr(Sig), L.ptr(MSig),
                                      0 if Sig is None else Sig.shape[1], mode, C.byref(val), L.ptr(g))
        if rc == L.NOT_SPD:
            return inf2
        self._check(rc, "nlml")
        return np.float64(val.value), (None if grad is None else g)

    def optimize(self, theta0, method="L-BFGS-B", grad="exact", **kw):
        """The reference's commented-out optimiser call (north/June1st.py:259-262:
        ``minimize(MLII, x0=[log l0, log sn0], method='CG', jac=True)``) against the device engine.
        ``grad='exact'`` (default) feeds the true derivative of the profiled nlML; ``grad='ref'`` reproduces the
        reference's MLII contract verbatim (its "gradient" is not the derivative, so CG stalls as in SURVEY App. C-7).
        Returns the scipy ``OptimizeResult``; afterwards the handle is fitted at ``exp(result.x)``."""
        from scipy.optimize import minimize

        if grad is None:                  # value only: scipy differences it numerically
            res = minimize(lambda th: float(self.nlml(th, grad=None)[0]), np.asarray(theta0, dtype=np.float64), method=method, jac=False, **kw)
        else:
            def fun(th):
                v, g = self.nlml(th, grad=grad)
                return float(v), np.asarray(g, dtype=np.float64)

            res = minimize(fun, np.asarray(theta0, dtype=np.float64), method=method, jac=True, **kw)
        if np.all(np.isfinite(res.x)):
            try:
                self.refit(float(np.exp(res.x[0])), float(np.exp(res.x[1])))
            except LinAlgError:
                pass
        return res

    # ---- state accessors -----------------------------------------------------------------------
    def _stat(self, name):
        v = C.c_double()
        self._check(self._lib.sigp_get_stat(self._h, name.encode(), C.byref(v)), "get_stat")
        return v.value

    @property
    def refine_residual_(self):
        """fp32 engine: max|y - K~ alpha~| / max|y| after the last fp64 refinement step of the last fit."""
        return self._stat("refine_residual")

    @property
    def matrix_bytes_(self):
        """Device bytes this handle holds in matrix / factor buffers."""
        return self._stat("matrix_bytes")

    @property
    def alpha_(self):
        """alpha = K^-1 y = A~/sigma_f (north/June1st.py:271), shape (n, 1)."""
        a = np.zeros(self.n)
        self._check(self._lib.sigp_get_alpha(self._h, L.ptr(a)), "get_alpha")
        return (a / self.sigma_f_).reshape(-1, 1)

    @property
    def L_tilde_(self):
        out = np.zeros((self.n, self.n))
        self._check(self._lib.sigp_get_matrix(self._h, 1, L.ptr(out), self.n), "get_matrix")
        return out

    @property
    def L_(self):
        """L = chol(K) = sqrt(sigma_f) L~ (north/June1st.py:270)."""
        return np.sqrt(self.sigma_f_) * self.L_tilde_

    def build(self, ell, sn_tilde):
        """K5 only: K~ (+ ride rows) into HBM; the factorisation is driven separately (potrf / DistributedGPR)."""
        if not self._has_data:
            raise RuntimeError("build: no data staged")
        if self.kernel == "netdiffusion":
            Sig = L.f64(self._sigma(float(ell)), 2)
            self._check(self._lib.sigp_kernel_build_from_sigma(self._h, L.ptr(Sig), Sig.shape[1], float(sn_tilde)), "kernel_build")
        else:
            self._check(self._lib.sigp_kernel_build(self._h, self._kid, float(ell), float(sn_tilde)), "kernel_build")
        self.ell_, self.sn_tilde_ = float(ell), float(sn_tilde)
        self._fitted = False

    def kernel_matrix(self, ell, sn_tilde):
        """K~ (lower triangle) as built on the device -- test/diagnostic accessor."""
        self.build(ell, sn_tilde)
        out = np.zeros((self.n, self.n))
        self._check(self._lib.sigp_get_matrix(self._h, 0, L.ptr(out), self.n), "get_matrix")
        self._fitted = False
        return out

    # ---- batches (retro loop :176-248, grid :210-211) ------------------------------------------
    def fit_batch(self, X, y, Xs, ell, sn_tilde, concurrency=2, group=8, M=None):
        """Independent fits.  X [B,n,d] (or [n,d] shared), y [B,n] (or [n]), Xs [B,m,d] (or [m,d] / None), ell [F],
        sn_tilde [F]; F fits, fit i uses data set i % B.
        Returns dict(sigma_f, nlml, info, sigma_n, mean [F,m], var [F,m]).

        RBF / Matern: lockstep groups on the blocked engine (data sets share (n, d, m)).
        Reference kernel: X / y / Xs may also be LISTS of arrays of different shapes (the retro years: n grows with the
        year); fits of order n <= 128 run one workgroup per fit in a single launch (``smallbatch.SmallBatch``), with
        Sigma~ = expm(l M) formed as this engine's ``expm`` option says; larger ones go one at a time through ``fit``."""
        if self.kernel == "netdiffusion":
            return self._fit_batch_netdiffusion(X, y, Xs, ell, sn_tilde, M)
        X = L.f64(X)
        shared = X.ndim == 2
        Xb = X[None] if shared else X
        B, n, d = Xb.shape
        yb = L.f64(np.asarray(y).reshape(B, n))
        m = 0
        Xsb = None
        if Xs is not None:
            Xsb = L.f64(Xs)
            Xsb = Xsb[None] if Xsb.ndim == 2 else Xsb
            if Xsb.shape[0] != B or Xsb.shape[2] != d:
                raise ValueError("Xs must be [B,m,d]")
            m = Xsb.shape[1]
        ell = L.f64(np.atleast_1d(ell), 1)
        sn = L.f64(np.atleast_1d(sn_tilde), 1)
        F = len(ell)
        if len(sn) != F:
            raise ValueError("ell and sn_tilde must have the same length")
        self._check(self._lib.sigp_batch_upload(self._h, B, L.ptr(Xb), n * d, L.ptr(yb), n, L.ptr(Xsb), m * d, n, d, m), "batch_upload")
        self._batch_m = m
        return self.run_batch(0, F, ell, sn, concurrency, group)

    def predict_batch(self, X, y, Xs, ell, sn_tilde, **kw):
        """(mean [F, m], var [F, m]) of a batch of independent fits at their test points -- the per-(region, year) outputs
        ``fmean`` / ``fvar`` of the retro loop (September1st_retro.py:236-242).  The test points ride along each factorisation,
        so this IS ``fit_batch``; it exists for callers that only want the predictions."""
        r = self.fit_batch(X, y, Xs, ell, sn_tilde, **kw)
        if np.any(r["info"] != 0):
            bad = int(np.flatnonzero(r["info"])[0])
            raise LinAlgError("Matrix is not positive definite (fit %d, pivot %d)" % (bad, r["info"][bad]), int(r["info"][bad]))
        return r["mean"], r["var"]

    def _fit_batch_netdiffusion(self, X, y, Xs, ell, sn_tilde, M=None):
        from .smallbatch import SmallBatch, NMAX, MMAX
        if isinstance(X, np.ndarray) and X.ndim == 2:
            X, y, Xs, M = [X], [y], [Xs], [M]
        B = len(X)
        Xs = [None] * B if Xs is None else list(Xs)
        if M is not None and isinstance(M, np.ndarray) and M.ndim == 2 and B > 1:
            raise ValueError("M must be None or a sequence of %d Laplacians (one per data set), not a single array" % B)
        M = [None] * B if M is None else list(M)
        if len(M) != B or len(Xs) != B:
            raise ValueError("X, Xs and M must have one entry per data set (%d)" % B)
        ell = np.atleast_1d(np.asarray(ell, dtype=np.float64)); sn = np.atleast_1d(np.asarray(sn_tilde, dtype=np.float64))
        F = len(ell)
        if len(sn) != F:
            raise ValueError("ell and sn_tilde must have the same length")
        mmax = max([0] + [np.atleast_2d(x).shape[0] for x in Xs if x is not None])
        res = dict(sigma_f=np.zeros(F), nlml=np.zeros(F), info=np.zeros(F, np.int64), sigma_n=np.zeros(F),
                   mean=np.full((F, mmax), np.nan), var=np.full((F, mmax), np.nan))
        small = [i for i in range(F) if np.asarray(X[i % B]).shape[0] <= NMAX and (Xs[i % B] is None or np.atleast_2d(Xs[i % B]).shape[0] <= MMAX)]
        if small:
            sb = SmallBatch(self)
            ids = {}
            for i in small:
                b = i % B
                if b not in ids:
                    ids[b] = sb.add_dataset(X[b], y[b], Xs[b], M[b])
                sb.add_fit(ids[b], ell[i], sn[i], expm=self._expm)
            r = sb.run()
            for k in ("sigma_f", "nlml", "info", "sigma_n"):
                res[k][small] = r[k]
            res["mean"][small, :r["mean"].shape[1]] = r["mean"]
            res["var"][small, :r["var"].shape[1]] = r["var"]
        small_set = set(small)
        for i in range(F):                                                # orders beyond one workgroup: the blocked engine, one fit at a time
            if i in small_set:
                continue
            b = i % B
            try:
                self.fit(X[b], y[b], ell[i], sn[i], M=M[b], Xs=Xs[b])
                res["sigma_f"][i], res["nlml"][i], res["sigma_n"][i] = self.sigma_f_, self.nlml_, self.sigma_n_
                if Xs[b] is not None:
                    mu, var = self.predict(Xs[b])
                    res["mean"][i, :len(mu)], res["var"][i, :len(var)] = mu, var
            except LinAlgError as e:
                res["sigma_f"][i] = res["nlml"][i] = res["sigma_n"][i] = np.inf
                res["info"][i] = e.info
        self._fitted = False
        return res

    def upload_batch(self, X, y, Xs, group=8, concurrency=1, M=None):
        """Stage data sets in HBM and allocate the lockstep slots without running any fit (bench warm-up).
        Reference kernel: X / y / Xs / M are sequences of ragged data sets (n <= 128 rows each: the retro years), kept as a
        ``SmallBatch`` whose data stay resident for ``nlml_batch`` / ``optimize_batch``."""
        if self.kernel == "netdiffusion":
            from .smallbatch import SmallBatch
            if isinstance(X, np.ndarray) and X.ndim == 2:
                X, y, Xs, M = [X], [y], [Xs], [M]
            B = len(X)
            Xs = [None] * B if Xs is None else list(Xs)
            M = [None] * B if M is None else list(M)
            if len(y) != B or len(Xs) != B or len(M) != B:
                raise ValueError("X, y, Xs and M must have one entry per data set (%d)" % B)
            sb = SmallBatch(self)
            self._small_ids = [sb.add_dataset(X[b], y[b], Xs[b], M[b]) for b in range(B)]
            self._small = sb
            return
        X = L.f64(X)
        Xb = X[None] if X.ndim == 2 else X
        B, n, d = Xb.shape
        yb = L.f64(np.asarray(y).reshape(B, n))
        m, Xsb = 0, None
        if Xs is not None:
            Xsb = L.f64(Xs)
            Xsb = Xsb[None] if Xsb.ndim == 2 else Xsb
            m = Xsb.shape[1]
        self._check(self._lib.sigp_batch_upload(self._h, B, L.ptr(Xb), n * d, L.ptr(yb), n, L.ptr(Xsb), m * d, n, d, m), "batch_upload")
        self._batch_m = m
        self._check(self._lib.sigp_batch_reserve(self._h, int(group), int(concurrency)), "batch_reserve")

    def run_batch(self, first, count, ell, sn_tilde, concurrency=2, group=8):
        """Run ``count`` fits on the data sets already resident in HBM (after fit_batch / upload).
        ``group`` fits are factorised in lockstep by each launch; ``concurrency`` groups are in flight."""
        self.set_option("group", group)
        ell = L.f64(np.atleast_1d(ell), 1)
        sn = L.f64(np.atleast_1d(sn_tilde), 1)
        out = np.zeros((count, 4))
        if len(ell) != count or len(sn) != count:
            raise ValueError("ell and sn_tilde must have `count` entries")
        mdim = getattr(self, "_batch_m", 0)
        mean = np.zeros((count, max(mdim, 1)))
        var = np.zeros((count, max(mdim, 1)))
        rc = self._lib.sigp_batch_run(self._h, int(first), int(count), self._kid, L.ptr(ell), L.ptr(sn), int(concurrency),
                                      L.ptr(out), L.ptr(mean) if mdim else None, L.ptr(var) if mdim else None)
        self._check(rc, "batch_run")
        self._fitted = False
        return dict(sigma_f=out[:, 0], nlml=out[:, 1], info=out[:, 2].astype(np.int64), sigma_n=out[:, 3],
                    mean=mean[:, :mdim], var=var[:, :mdim])

    def nlml_batch(self, theta, first=0, grad="exact", group=8, expm="eigh", sets=None):
        """``MLII`` for many (data set, theta) pairs in one device call on the data sets staged by ``upload_batch`` / ``fit_batch``:
        theta [F, 2] = (log l, log sn~), pair i uses data set (first + i) % B.  Returns (nlml [F], grad [F, 2] or None), +inf where K~
        is not SPD or exp(theta) overflows (north/June1st.py:254-256).  This is what an optimiser over all retrospective years
        evaluates per iteration (the reference's commented-out call, :259-262).

        RBF / Matern: lockstep groups on the blocked engine; grad = 'exact' (the derivative of the profiled nlML) or None.
        Reference kernel (n <= 128 per data set): one workgroup per pair, one launch (``sigp_small_run_grad``); grad = 'ref'
        reproduces the reference's own formulae (:248-252), 'exact' the true derivative.  ``expm`` = 'eigh' (default: one
        eigendecomposition of M per data set, nothing but theta crosses the bus per call) or 'pade' (scipy's expm per pair on
        the host, the reference's own numbers at extreme l); ``sets`` [F] names the data set of every pair explicitly."""
        theta = L.f64(np.atleast_2d(theta), 2)
        if theta.shape[1] != 2:
            raise ValueError("theta must be [F, 2]")
        F = theta.shape[0]
        if self.kernel == "netdiffusion":
            if grad not in (None, "ref", "exact"):
                raise ValueError("grad must be None, 'ref' or 'exact'")
            sb = getattr(self, "_small", None)
            if sb is None:
                raise RuntimeError("nlml_batch: stage the data sets with upload_batch(X_list, y_list, Xs_list, M=M_list) first")
            B = len(self._small_ids)
            with np.errstate(over="ignore"):
                ell, sn = np.exp(theta[:, 0]), np.exp(theta[:, 1])
            ok = np.isfinite(ell) & np.isfinite(sn) & (ell > 0)
            val = np.full(F, np.inf)
            g = np.full((F, 2), np.inf)
            sb.clear_fits()
            run = []
            for i in np.flatnonzero(ok):
                try:
                    sb.add_fit(self._small_ids[(first + i) % B if sets is None else int(sets[i])], ell[i], sn[i], expm=expm)
                    run.append(i)
                except (FloatingPointError, OverflowError, ValueError):       # scipy's expm overflowed: the except branch of :254-256
                    pass
            if run:
                r = sb.run(grad=grad is not None)
                val[run] = r["nlml"]
                if grad is not None:
                    g[run] = r["grad_ref" if grad == "ref" else "grad_exact"]
            return val, (g if grad is not None else None)
        if grad not in (None, "exact"):
            raise ValueError("grad must be None or 'exact'")
        if sets is not None:
            raise ValueError("sets= is for the reference kernel's ragged data sets; the lockstep groups pair theta i with data set (first + i) % B")
        self.set_option("group", group)
        val = np.zeros(F)
        g = np.zeros((F, 2))
        self._check(self._lib.sigp_nlml_grad_batch(self._h, int(first), F, self._kid, L.ptr(theta), 0 if grad is None else 2, L.ptr(val),
                                                   L.ptr(g) if grad is not None else None), "nlml_batch")
        self._fitted = False
        return val, (g if grad is not None else None)

    def optimize_batch(self, X, y, theta0, group=8, maxiter=50, gtol=1e-5, ftol=1e-10, max_step=2.0, M=None, expm="eigh", method=None):
        """The reference's commented-out ``minimize(MLII, x0, method='CG', jac=True)`` (north/June1st.py:259-262) for EVERY data set
        of a retrospective run at once: X [B, n, d], y [B, n], theta0 [B, 2] (or [2]) -> dict(x [B, 2], fun [B], nit [B],
        converged [B], nfev = device calls).  The state of every data set lives on the host and each round is ONE device call for all
        unfinished data sets: the years advance together whatever their individual line searches do (``optim.py``).

        RBF / Matern (``method='bfgs'``): BFGS on the 2-vector (log l, log sn~) per data set with Armijo backtracking, one trial point
        per data set and round in lockstep groups (``nlml_batch``).
        Reference kernel (``method='newton'``): X / y (/ M) are sequences of ragged data sets (the 3 regions x years of
        September1st_retro.py:176-180); a round is ONE launch of one workgroup per point, value + exact gradient formed in LDS, and since
        extra points are free there each round carries several step lengths and their finite-difference neighbours: a modified-Newton
        iteration with the line search inside the launch."""
        from .optim import bfgs_lockstep, newton_lockstep
        if self.kernel == "netdiffusion":
            self.upload_batch(X, y, None, M=M)
            B = len(self._small_ids)
            th0 = np.broadcast_to(np.asarray(theta0, dtype=np.float64), (B, 2))
            if (method or "newton") == "newton":
                return newton_lockstep(lambda t, own: self.nlml_batch(t, grad="exact", expm=expm, sets=own), th0, maxiter=maxiter, gtol=gtol,
                                       ftol=min(ftol, 1e-12), max_step=max_step)
            return bfgs_lockstep(lambda t: self.nlml_batch(t, grad="exact", expm=expm), th0, maxiter=maxiter, gtol=gtol, ftol=ftol, max_step=max_step)
        if (method or "bfgs") != "bfgs":
            raise ValueError("method='newton' needs the one-workgroup-per-point kernel (reference kernel); RBF / Matern take 'bfgs'")
        X = L.f64(X, 3)
        B = X.shape[0]
        self.upload_batch(X, y, None, group=group, concurrency=1)
        th0 = np.broadcast_to(np.asarray(theta0, dtype=np.float64), (B, 2))
        return bfgs_lockstep(lambda t: self.nlml_batch(t, grad="exact", group=group), th0, maxiter=maxiter, gtol=gtol, ftol=ftol, max_step=max_step)

    def nlml_grid(self, X, y, ells, sns, concurrency=2, group=8, M=None):
        """nlML on the (l, sn~) grid for one data set -- the offline 20x20 search implied by
        north/June1st.py:210-211 (``ells = LGRID, sns = SGRID`` with the reference kernel reproduces it: one launch,
        one workgroup per grid point).  Returns [len(ells), len(sns)], +inf where K~ is not SPD."""
        ells = np.asarray(ells, dtype=np.float64).reshape(-1)
        sns = np.asarray(sns, dtype=np.float64).reshape(-1)
        E, S = np.meshgrid(ells, sns, indexing="ij")
        if self.kernel == "netdiffusion":
            r = self.fit_batch(X, y, None, E.reshape(-1), S.reshape(-1), M=M)
        else:
            r = self.fit_batch(X, y, None, E.reshape(-1), S.reshape(-1), concurrency=concurrency, group=group)
        return r["nlml"].reshape(len(ells), len(sns))

    # ---- the feature pipeline's correlation threshold on the device (networks.Network.tau(engine=gp)) --------------
    def corr_tau(self, series, dof, significance):
        """Cell-to-cell correlation matrix of ``series`` [N, T] and the mean of its significantly positive entries
        (behaviour of ComplexNetworks.py:31-47): returns (R [N, N] with a NaN diagonal, tau)."""
        from scipy import stats
        series = L.f64(series, 2)
        N, T = series.shape
        t_c = float(stats.t.isf(significance, dof))
        r_crit = t_c / np.sqrt(dof + t_c * t_c)
        R = np.empty((N, N))
        s, c = C.c_double(), C.c_double()
        self._check(self._lib.sigp_corr_tau(self._h, L.ptr(series), N, T, T, r_crit, L.ptr(R), N, C.byref(s), C.byref(c)), "corr_tau")
        with np.errstate(invalid="ignore", divide="ignore"):
            return R, np.float64(s.value) / np.float64(c.value)

    def area_sums(self, data, weight, label, nareas):
        """Per-area weighted sums of ``data`` [X, Y, T] (networks.Network.intra_links(engine=gp)): out [nareas, T]."""
        data = L.f64(data, 3)
        X, Y, T = data.shape
        w = L.f64(np.broadcast_to(weight, (X, Y)), 2)
        lab = np.ascontiguousarray(label, dtype=np.int32).reshape(X * Y)
        out = np.empty((int(nareas), T))
        self._check(self._lib.sigp_area_sums(self._h, L.ptr(data), X * Y, T, L.ptr(w), L.iptr(lab), int(nareas), L.ptr(out)), "area_sums")
        return out

    def detrend_cuts(self, data, cut_lens):
        """Per-pixel line removal of ``data`` [X, Y, T] over its first ``cut_lens[c]`` steps, every cut in one launch
        (callers.detrend(engine=gp)).  Returns ([dt [X, Y, n_c] ...], [trend [X, Y, 2] ...])."""
        data = L.f64(data, 3)
        X, Y, T = data.shape
        cuts = np.ascontiguousarray(cut_lens, dtype=np.int64)
        dt = np.empty(int(X * Y * cuts.sum()))
        tr = np.empty((len(cuts), X, Y, 2))
        self._check(self._lib.sigp_detrend(self._h, L.ptr(data), X * Y, T, len(cuts), L.iptr(cuts), L.ptr(dt), L.ptr(tr)), "detrend")
        outs, o = [], 0
        for n in cuts:
            outs.append(dt[o:o + X * Y * n].reshape(X, Y, int(n)))
            o += X * Y * int(n)
        return outs, [tr[c] for c in range(len(cuts))]

    # ---- measurement ---------------------------------------------------------------------------
    def profile(self, enable=True, classes=None):
        """Bracket kernel launches with HIP events (all classes, or only the named ones, e.g. ["syrk128"])."""
        code = int(bool(enable))
        if enable and classes:
            code = 0
            for c in classes:
                code |= 1 << (8 + L.KCLASS[c])
        self._check(self._lib.sigp_profile(self._h, code), "profile")

    def profile_reset(self):
        self._check(self._lib.sigp_profile_reset(self._h), "profile_reset")

    def synchronize(self):
        """Wait for everything the handle's device has been given (sigp_synchronize): the bracket of a timed region, on the
        HIP runtime the library itself links."""
        self._check(self._lib.sigp_synchronize(self._h), "synchronize")

    def profile_get(self):
        """{kernel class: dict(ms, launches, flops, bytes)} accumulated since the last reset."""
        out = {}
        for name, k in L.KCLASS.items():
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            nl = C.c_int64()
            self._check(self._lib.sigp_profile_get(self._h, k, C.byref(ms), C.byref(nl), C.byref(fl), C.byref(by)), "profile_get")
            out[name] = dict(ms=ms.value, launches=nl.value, flops=fl.value, bytes=by.value)
        return out
