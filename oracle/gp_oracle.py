"""CPU oracle for the GPR hot path of William-gregory/SeaIceExtentForecasting.

TEST INFRASTRUCTURE ONLY.  Nothing under ``seaiceextentforecasting_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / the timed CPU baseline, never as the product path.

What it is: a NumPy/SciPy restatement of the inline Gaussian-process block that every forecast
script of the reference runs (byte-identical in all 14 scripts):

    /root/reference/north/June1st.py:214-233   feature selection, design matrix, graph Laplacian
    /root/reference/north/June1st.py:235-257   MLII(theta) -> (nlML, "gradient")
    /root/reference/north/June1st.py:263-277   fit (two-stage Cholesky, profiled sigma_f) + predict
    /root/reference/north/retrospective_forecasts/September1st_retro.py:171-249   retro loop

Parity status
-------------
* network-diffusion kernel (the reference's own ``K = X expm(l M) X^T + sn I``): PINNED.  The
  restatement is checked against captures of the reference's own ``forecast()`` executed in the
  authoring container on synthetic inputs (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
  ``tests/test_oracle_golden.py``), NumPy 2.2.6 / SciPy 1.15.3.
* RBF and Matern-5/2 kernels (BASELINE.json configs): the reference has no such kernels, so
  parity for the *kernel function itself* is "parity unpinned" by the reference; the fit/solve/
  predict skeleton around them is the pinned one above, and the kernel functions are cross-checked
  against scikit-learn's RBF / Matern(nu=2.5) in ``tests/test_oracle_golden.py``.

Two flavours of the linear algebra are provided:
  ref_idiom=True   call-for-call the reference's sequence (expm twice, cholesky twice,
                   ``np.linalg.solve`` on triangular factors: June1st.py:264-274)
  ref_idiom=False  best-practice CPU path (one Cholesky, ``solve_triangular``)
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm, solve_triangular
from scipy.stats import pearsonr

# --------------------------------------------------------------------------------------------
# a11: hyperparameter tables  (north/June1st.py:210-211 and the 13 sibling scripts)
# --------------------------------------------------------------------------------------------
LGRID = np.logspace(-7, 2, 20)   # l axis      (north/June1st.py:210)
SGRID = np.logspace(-3, 9, 20)   # sn~ axis    (north/June1st.py:211)

#: script key -> dict(rule, standardise, ell[3], sn[3], drop_first, regions)
SCRIPT_TABLE = {
    # north/June1st.py:209-211, 217-227
    "north_June": dict(rule="sic_pos_sst_neg", standardise=True, drop_first=False,
                       ell=[LGRID[16], LGRID[14], LGRID[12]], sn=[SGRID[1], SGRID[4], SGRID[6]],
                       regions=["Pan-Arctic", "Beaufort", "Chukchi"]),
    # north/July1st.py:168-179
    "north_July": dict(rule="pos", standardise=False, drop_first=False,
                       ell=[LGRID[11], LGRID[0], 3.125433e+10], sn=[SGRID[4], SGRID[15], 40221.26298973],
                       regions=["Pan-Arctic", "Beaufort", "Chukchi"]),
    # north/August1st.py:168-182
    "north_August": dict(rule="all_then_pos_p", pthr=0.08, standardise=False, drop_first=False,
                         ell=[LGRID[9], LGRID[7], LGRID[3]], sn=[SGRID[4], SGRID[13], SGRID[13]],
                         regions=["Pan-Arctic", "Beaufort", "Chukchi"]),
    # north/September1st.py:169-183
    "north_September": dict(rule="all_then_pos_p", pthr=0.05, standardise=False, drop_first=False,
                            ell=[LGRID[8], LGRID[9], LGRID[3]], sn=[SGRID[6], SGRID[3], SGRID[13]],
                            regions=["Pan-Arctic", "Beaufort", "Chukchi"]),
    # south/December1st.py:161-171
    "south_December": dict(rule="pos", standardise=False, drop_first=True,
                           ell=[LGRID[4], LGRID[9], LGRID[2]], sn=[SGRID[13], SGRID[4], SGRID[13]],
                           regions=["Pan-Antarctic", "Ross", "Weddell"]),
    # south/January1st.py:162-175
    "south_January": dict(rule="all_then_pos_p", pthr=0.08, standardise=False, drop_first=True,
                          ell=[LGRID[2], LGRID[1], LGRID[3]], sn=[SGRID[14], SGRID[14], SGRID[14]],
                          regions=["Pan-Antarctic", "Ross", "Weddell"]),
    # south/February1st.py:161-174
    "south_February": dict(rule="all_then_pos_p", pthr=0.05, standardise=False, drop_first=False,
                           ell=[LGRID[16], LGRID[5], LGRID[3]], sn=[SGRID[0], SGRID[11], SGRID[13]],
                           regions=["Pan-Antarctic", "Ross", "Weddell"]),
}


# --------------------------------------------------------------------------------------------
# a1: feature selection + design matrix   (north/June1st.py:214-229 and variants)
# --------------------------------------------------------------------------------------------
def select_features(y, sic_anoms, sst_anoms=None, *, rule, k, pthr=None):
    """Return the list of (n+1)-vectors the reference appends to ``X`` for region index ``k``.

    rule 'sic_pos_sst_neg' : north/June1st.py:217-224   (SIC r>0; SST r<0, sign flipped)
    rule 'pos'             : north/July1st.py:176-179, south/December1st.py:168-171
    rule 'all_then_pos_p'  : north/August1st.py:176-182 (k==0: every area; k>0: r>0 & p/2<pthr)
    """
    y0 = np.asarray(y)[:, 0]
    feats = []
    for area in sic_anoms:                                   # dict order, as the reference iterates
        r, p = pearsonr(y0, sic_anoms[area][:-1])
        if rule in ("pos", "sic_pos_sst_neg"):
            if r > 0:
                feats.append(sic_anoms[area])
        elif rule == "all_then_pos_p":
            if k == 0:
                feats.append(sic_anoms[area])
            elif (r > 0) & (p / 2 < pthr):
                feats.append(sic_anoms[area])
        else:
            raise ValueError(rule)
    if rule == "sic_pos_sst_neg":
        for area in sst_anoms:
            r, p = pearsonr(y0, sst_anoms[area][:-1])
            if r < 0:
                feats.append(-sst_anoms[area])
    return feats


def design_matrix(feats, standardise):
    """north/June1st.py:226-229.  Returns (X [n,N], Xs [1,N]).

    Standardisation (June only) uses mean/std over all n+1 rows *including the test row*, ddof=0.
    """
    X = np.asarray(feats).T                    # (n+1) x N (a transposed view, as in the reference)
    if standardise:
        X = (X - np.mean(X, 0)) / np.std(X, 0)
    Xs = np.asarray([X[-1, :]])
    X = X[:-1, :]
    return X, Xs


# --------------------------------------------------------------------------------------------
# a2: graph Laplacian   (north/June1st.py:231-233)
# --------------------------------------------------------------------------------------------
def laplacian_M(X):
    M = np.abs(np.cov(X, rowvar=False, bias=True))
    M = np.atleast_2d(M)
    np.fill_diagonal(M, 0)
    np.fill_diagonal(M, -np.sum(M, axis=0))
    return M


# --------------------------------------------------------------------------------------------
# a3/a4: covariance functions
# --------------------------------------------------------------------------------------------
def sigma_tilde(M, ell):
    """a3: S~ = expm(l*M)   (north/June1st.py:264)."""
    return expm(ell * M)


def sqdist(A, B):
    """Squared euclidean distances by direct differences (no GEMM-form cancellation)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    out = np.zeros((A.shape[0], B.shape[0]))
    for p in range(A.shape[1]):
        diff = A[:, p][:, None] - B[:, p][None, :]
        out += diff * diff
    return out


def cov_unit(kind, A, B, ell, Sigma=None):
    """Unit-signal-variance covariance k~(A,B) (no noise term).

    'netdiffusion' : A Sigma B^T with Sigma = expm(l M)  (north/June1st.py:265, 272) -- reference kernel
    'rbf'          : exp(-|a-b|^2 / (2 l^2))             -- added by BASELINE.json configs
    'matern52'     : (1 + s + s^2/3) exp(-s), s = sqrt(5) |a-b| / l
    """
    if kind == "netdiffusion":
        return np.linalg.multi_dot([A, Sigma, B.T])
    D2 = sqdist(A, B)
    if kind == "rbf":
        return np.exp(-0.5 * D2 / (ell * ell))
    if kind == "matern52":
        s = np.sqrt(5.0 * D2) / ell
        return (1.0 + s + s * s / 3.0) * np.exp(-s)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------------
# a4-a8: fit + predict   (north/June1st.py:263-277)
# --------------------------------------------------------------------------------------------
def fit_predict(X, y, Xs, ell, sn_tilde, *, kind="netdiffusion", M=None, ref_idiom=True):
    """One GP fit + prediction.  Returns a dict with every intermediate of the reference block.

    y is (n,1) (north/June1st.py:214); Xs is (m,N).  ``fvar`` includes the noise variance sigma_n
    (north/June1st.py:273, 277).  For m>1 the reference formulae are applied per test point
    (fmean[m], fvar[m] = diagonal of the reference's 1x1 expressions).
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    Xs = np.atleast_2d(np.asarray(Xs, dtype=np.float64))
    n = len(y)
    out = {}
    St = None
    if kind == "netdiffusion":
        if M is None:
            M = laplacian_M(X)
        St = expm(ell * M)                                                     # :264
    Kt = cov_unit(kind, X, X, ell, St) + np.eye(n) * sn_tilde                  # :265
    L_tilde = np.linalg.cholesky(Kt)                                           # :265
    if ref_idiom:
        A_tilde = np.linalg.solve(L_tilde.T, np.linalg.solve(L_tilde, y))      # :266
    else:
        A_tilde = solve_triangular(L_tilde.T, solve_triangular(L_tilde, y, lower=True), lower=False)
    sf = (np.dot(y.T, A_tilde) / n)[0][0]                                      # :267
    sn = sf * sn_tilde                                                         # :268
    if ref_idiom:
        S = sf * expm(ell * M) if kind == "netdiffusion" else None             # :269
        K = (np.linalg.multi_dot([X, S, X.T]) if kind == "netdiffusion"
             else sf * cov_unit(kind, X, X, ell)) + np.eye(n) * sn             # :270
        L = np.linalg.cholesky(K)                                              # :270
        alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))                    # :271
    else:
        S = sf * St if kind == "netdiffusion" else None
        L = np.sqrt(sf) * L_tilde            # K = sf*K~  =>  L = sqrt(sf) L~   
        alpha = A_tilde / sf
    KXXs = sf * cov_unit(kind, X, Xs, ell, St) if not (ref_idiom and kind == "netdiffusion") \
        else np.linalg.multi_dot([X, S, Xs.T])                                 # :272
    if kind == "netdiffusion":
        KXsXs_full = np.linalg.multi_dot([Xs, S, Xs.T]) + sn                   # :273
        kss = np.diag(KXsXs_full).copy()
    else:
        kss = np.full(Xs.shape[0], sf * 1.0 + sn)   # k~(x,x)=1 for rbf/matern
    if ref_idiom:
        v = np.linalg.solve(L, KXXs)                                           # :274
    else:
        v = solve_triangular(L, KXXs, lower=True)
    fmean = np.dot(KXXs.T, alpha)[:, 0]                                        # :276
    fvar = kss - np.sum(v * v, axis=0)                                         # :277
    nlml = (np.dot(y.T, alpha) / 2 + np.log(L.diagonal()).sum() + n * np.log(2 * np.pi) / 2)[0][0]  # :246
    out.update(M=M, Sigma_tilde=St, K_tilde=Kt, L_tilde=L_tilde, A_tilde=A_tilde, sigma_f=sf,
               sigma_n=sn, Sigma=S, L=L, alpha=alpha, KXXs=KXXs, kss=kss, v=v,
               fmean=fmean, fvar=fvar, nlml=nlml)
    return out


def _tri_solve_blocked(L, B, trans=False, bs=2048):
    """L x = B (or L^T x = B) for a large lower-triangular L by block substitution: diagonal blocks through LAPACK trtrs, the
    rest through NumPy matmul (NumPy's BLAS takes 64-bit sizes; SciPy's LP64 LAPACK segfaults on a 32768 x 32768 matrix)."""
    n = L.shape[0]
    X = np.array(B, dtype=np.float64, copy=True)
    starts = list(range(0, n, bs))
    if not trans:
        for i0 in starts:
            i1 = min(n, i0 + bs)
            X[i0:i1] = solve_triangular(L[i0:i1, i0:i1], X[i0:i1], lower=True, check_finite=False)
            if i1 < n:
                X[i1:] -= L[i1:, i0:i1] @ X[i0:i1]
    else:
        for i0 in reversed(starts):
            i1 = min(n, i0 + bs)
            X[i0:i1] = solve_triangular(L[i0:i1, i0:i1], X[i0:i1], lower=True, trans="T", check_finite=False)
            if i0 > 0:
                X[:i0] -= L[i0:i1, :i0].T @ X[i0:i1]
    return X


def _cholesky_blocked(K, bs=4096):
    """In-place lower Cholesky of a large SPD matrix by the blocked right-looking algorithm LAPACK's potrf uses, spelled out
    in NumPy / SciPy calls on blocks (this image's OpenBLAS 0.3.29 segfaults in potrf at n = 32768, in NumPy and SciPy alike):
    diagonal block through ``np.linalg.cholesky``, the rows below through trsm, the trailing update through matmul, one block
    column at a time so that temporaries stay at n x bs.  The strictly upper part is zeroed, as np.linalg.cholesky does."""
    n = K.shape[0]
    for j0 in range(0, n, bs):
        j1 = min(n, j0 + bs)
        K[j0:j1, j0:j1] = np.linalg.cholesky(K[j0:j1, j0:j1])
        if j1 < n:
            # L21 = A21 L11^-T  <=>  L11 L21^T = A21^T
            K[j1:, j0:j1] = solve_triangular(K[j0:j1, j0:j1], K[j1:, j0:j1].T, lower=True, check_finite=False).T
            for c0 in range(j1, n, bs):
                c1 = min(n, c0 + bs)
                K[c0:, c0:c1] -= K[c0:, j0:j1] @ K[c0:c1, j0:j1].T
        K[j0:j1, j1:] = 0.0
    return K


def fit_predict_lean(X, y, Xs, ell, sn_tilde, *, kind="rbf", row_block=1024, threads=16):
    """The ``ref_idiom=False`` statements of :func:`fit_predict` (north/June1st.py:265-277, :246) for RBF / Matern at sizes
    where its n x n temporaries do not fit: K~ is formed in row blocks (same arithmetic per entry: squared distances by
    direct differences, then the covariance function), factored by ``np.linalg.cholesky`` as in the reference (:265; beyond
    n = 16384 by the same blocked algorithm spelled out on blocks, ``_cholesky_blocked``), and only the scalars of the block
    are returned: fmean, fvar, sigma_f, sigma_n, nlml, A_tilde.  At most two n x n float64 arrays at peak (16 GiB at n = 32768).  Checked against :func:`fit_predict` in tests/test_oracle_golden.py."""
    from concurrent.futures import ThreadPoolExecutor
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    Xs = np.atleast_2d(np.asarray(Xs, dtype=np.float64))
    n = len(y)
    Kt = np.empty((n, n))

    def rows(i0):
        i1 = min(n, i0 + row_block)
        blk = cov_unit(kind, X[i0:i1], X, ell)                  # :265 (RBF / Matern in place of X Sigma X^T)
        idx = np.arange(i0, i1)
        blk[idx - i0, idx] += sn_tilde
        Kt[i0:i1, :] = blk

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(rows, range(0, n, row_block)))
    L_tilde = _cholesky_blocked(Kt) if n > 16384 else np.linalg.cholesky(Kt)             # :265
    del Kt
    A_tilde = _tri_solve_blocked(L_tilde, _tri_solve_blocked(L_tilde, y), trans=True)    # :266
    sf = float(y[:, 0] @ A_tilde[:, 0]) / n                                              # :267
    sn = sf * sn_tilde                                                                   # :268
    ks = cov_unit(kind, X, Xs, ell)                                                      # :272 in unit signal variance
    v = _tri_solve_blocked(L_tilde, ks)                                                  # :274 (L = sqrt(sf) L~)
    fmean = ks.T @ A_tilde[:, 0]                                                         # :276 (k* alpha = k~* A~)
    fvar = sf * (1.0 + sn_tilde - np.sum(v * v, axis=0))                                 # :273, :277
    nlml = 0.5 * n + np.log(L_tilde.diagonal()).sum() + 0.5 * n * np.log(sf) + 0.5 * n * np.log(2 * np.pi)   # :246, y^T alpha = n
    return dict(fmean=fmean, fvar=fvar, sigma_f=sf, sigma_n=sn, nlml=nlml, A_tilde=A_tilde)


# --------------------------------------------------------------------------------------------
# a9: MLII   (north/June1st.py:235-257)
# --------------------------------------------------------------------------------------------
def mlii(theta, X, y, *, kind="netdiffusion", M=None, grad="ref"):
    """Negative log marginal likelihood + the reference's 2-vector "gradient".

    grad='ref' reproduces the reference formulae (north/June1st.py:248-252) -- these are NOT the
    derivative of nlML; grad='exact' is the analytic derivative of the profiled
    nlML w.r.t. (log l, log sn~).  Failure (non-SPD, overflow) -> (inf, [inf, inf]) (:254-256).
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    n = len(y)
    ell = np.exp(theta[0])
    snt = np.exp(theta[1])                                                     # :236
    try:
        with np.errstate(over="raise", invalid="raise"):
            if kind == "netdiffusion":
                if M is None:
                    M = laplacian_M(X)
                St = expm(ell * M)                                             # :238
            else:
                St = None
            Ku = cov_unit(kind, X, X, ell, St)
            L_tilde = np.linalg.cholesky(Ku + np.eye(n) * snt)                 # :239
            A_tilde = np.linalg.solve(L_tilde.T, np.linalg.solve(L_tilde, y))  # :240
            sf = (np.dot(y.T, A_tilde) / n)[0][0]                              # :241
            sn = sf * snt                                                      # :242
            if kind == "netdiffusion":
                S = sf * expm(ell * M)                                         # :243
                K = np.linalg.multi_dot([X, S, X.T]) + np.eye(n) * sn          # :244
            else:
                K = sf * Ku + np.eye(n) * sn
            L = np.linalg.cholesky(K)                                          # :244
            alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))                # :245
            nlml = (np.dot(y.T, alpha) / 2 + np.log(L.diagonal()).sum() + n * np.log(2 * np.pi) / 2)  # :246
            if grad == "ref":
                if kind != "netdiffusion":
                    raise ValueError("grad='ref' is defined for the reference kernel only")
                dKdl = np.linalg.multi_dot([X, np.dot(M, S), X.T]) + np.eye(n) * sn      # :248
                dKds = np.linalg.multi_dot([X, S, X.T]) + np.eye(n) * sf                 # :249
                g1 = (np.trace(np.linalg.solve(L.T, np.linalg.solve(L, dKdl))) / 2
                      - np.linalg.multi_dot([alpha.T, dKdl, alpha]) / 2)[0][0]           # :251
                g2 = (np.trace(np.linalg.solve(L.T, np.linalg.solve(L, dKds))) / 2
                      - np.linalg.multi_dot([alpha.T, dKds, alpha]) / 2)[0][0]           # :252
            elif grad == "exact":
                # profiled objective: f(theta) = n/2 + 1/2 log|K~| + n/2 log sf + n/2 log 2pi,
                # sf = y^T K~^-1 y / n.  df = 1/2 tr(K~^-1 dK~) - 1/(2 sf) a~^T dK~ a~
                if kind == "netdiffusion":
                    dK1 = ell * np.linalg.multi_dot([X, np.dot(M, St), X.T])
                elif kind == "rbf":
                    D2 = sqdist(X, X)
                    dK1 = Ku * D2 / (ell * ell)
                else:  # matern52: dk/dlog l = (s^2/3)(1+s) exp(-s)
                    s = np.sqrt(5.0 * sqdist(X, X)) / ell
                    dK1 = (s * s / 3.0) * (1.0 + s) * np.exp(-s)
                dK2 = snt * np.eye(n)
                Kinv = solve_triangular(L_tilde.T, solve_triangular(L_tilde, np.eye(n), lower=True), lower=False)
                g = []
                for dK in (dK1, dK2):
                    g.append(0.5 * np.sum(Kinv * dK) - 0.5 / sf * (A_tilde.T @ dK @ A_tilde)[0][0])
                g1, g2 = g
            else:
                raise ValueError(grad)
    except (np.linalg.LinAlgError, ValueError, OverflowError, FloatingPointError):
        return np.inf, np.asarray([np.inf, np.inf])                            # :254-256
    return np.squeeze(nlml), np.asarray([g1, g2])                              # :257


# --------------------------------------------------------------------------------------------
# a10: retrospective batch loop   (north/retrospective_forecasts/September1st_retro.py:171-249)
# --------------------------------------------------------------------------------------------
def retro_forecast(script, SIC, SIEs_dt, SIEs_trend, fmin, fmax, SST=None, *, ref_idiom=True, fit=None):
    """Restatement of the retro ``forecast(fmin,fmax)``: 3 regions x years of a1-a8, ``.round(3)``.

    ``fit`` may be supplied to swap the a4-a8 engine (signature of :func:`fit_predict`); the tests use
    this to run the HIP engine through the same host loop.
    Index conventions: y row = year-(fmin-1)-1, columns range(year-1979) (September1st_retro.py:181);
    south Dec/Jan drop the first target and use the previous year's networks
    (south/retrospective_forecasts/January1st_retro.py:173-176).
    """
    tab = SCRIPT_TABLE[script]
    fit = fit or (lambda X, y, Xs, ell, sn, M: fit_predict(X, y, Xs, ell, sn, M=M, ref_idiom=ref_idiom))
    GPR = {}
    for k, region in enumerate(tab["regions"]):
        ny = fmax - fmin + 1
        fmean = np.zeros(ny)
        fvar = np.zeros(ny)
        fmean_rt = np.zeros(ny)
        for year in range(fmin, fmax + 1):
            row = year - (fmin - 1) - 1
            cols = range(1, year - 1979) if tab["drop_first"] else range(year - 1979)
            y = np.asarray([SIEs_dt[region][row, cols]]).T
            key = "anoms_" + str(year - 1 if tab["drop_first"] else year)
            feats = select_features(y, SIC[key], SST[key] if SST is not None else None,
                                    rule=tab["rule"], k=k, pthr=tab.get("pthr"))
            X, Xs = design_matrix(feats, tab["standardise"])
            M = laplacian_M(X)
            r = fit(X, y, Xs, tab["ell"][k], tab["sn"][k], M)
            fmean[year - fmin] = np.round(r["fmean"][0], 3)                                     # :241
            fvar[year - fmin] = np.round(r["fvar"][0], 3)                                       # :242
            lineT = (np.arange(year - 1979 + 1) * SIEs_trend[region][row, 0]) + SIEs_trend[region][row, 1]
            fmean_rt[year - fmin] = (fmean[year - fmin] + lineT[-1]).round(3)                   # :244
        GPR[region + "_fmean"] = fmean
        GPR[region + "_fvar"] = fvar
        GPR[region + "_fmean_rt"] = fmean_rt
    return GPR


def operational_forecast(script, SIC, SIEs_dt, SIEs_trend, ymax, SST=None, *, ref_idiom=True, fit=None):
    """Restatement of the operational ``forecast(ymax)`` (north/June1st.py:208-279), unrounded."""
    tab = SCRIPT_TABLE[script]
    fit = fit or (lambda X, y, Xs, ell, sn, M: fit_predict(X, y, Xs, ell, sn, M=M, ref_idiom=ref_idiom))
    out = {}
    for k, region in enumerate(tab["regions"]):
        yv = SIEs_dt[region][1:] if tab["drop_first"] else SIEs_dt[region]
        y = np.asarray([yv]).T
        feats = select_features(y, SIC["anoms"], SST["anoms"] if SST is not None else None,
                                rule=tab["rule"], k=k, pthr=tab.get("pthr"))
        X, Xs = design_matrix(feats, tab["standardise"])
        M = laplacian_M(X)
        r = fit(X, y, Xs, tab["ell"][k], tab["sn"][k], M)
        lineT = (np.arange(ymax - 1979 + 1) * SIEs_trend[region][0]) + SIEs_trend[region][1]   # :278
        out[region] = dict(fmean=r["fmean"][0], fvar=r["fvar"][0], fmean_rt=r["fmean"][0] + lineT[-1])
    return out


# --------------------------------------------------------------------------------------------
# synthetic workloads of BASELINE.json / SURVEY 8(d)
# --------------------------------------------------------------------------------------------
def synthetic_problem(n, d, seed, m=1):
    """SURVEY 8(d) synthetic inputs: X~N(0,1), y = sin(Xw) + 0.1 eps, Xs~N(0,1); l = sqrt(d)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal(n)
    Xs = rng.standard_normal((m, d))
    return X, y, Xs
