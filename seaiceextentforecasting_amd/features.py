"""Host side of the hot path that stays on the CPU: feature selection,
design matrix, graph Laplacian, Sigma~ = expm(l M), and the per-script hyper-parameter tables.

These feed the HIP engine (gpr.GPR); they are tiny (n <= ~45 years, N <= ~200 areas) and are part of
the kept ComplexNetworks feature pipeline, so they are vectorised NumPy rather than kernels.
"""
import numpy as np
from scipy import special
from scipy.linalg import expm

# a11 -- grids and per-region picks (north/June1st.py:210-211 and siblings) ------------
LGRID = np.logspace(-7, 2, 20)
SGRID = np.logspace(-3, 9, 20)
_N = ["Pan-Arctic", "Beaufort", "Chukchi"]
_S = ["Pan-Antarctic", "Ross", "Weddell"]


def _row(rule, l_idx, s_idx, regions, standardise=False, drop_first=False, pthr=None):
    pick = lambda grid, v: [grid[i] if isinstance(i, (int, np.integer)) else float(i) for i in v]
    return dict(rule=rule, standardise=standardise, drop_first=drop_first, pthr=pthr, regions=regions,
                ell=pick(LGRID, l_idx), sn=pick(SGRID, s_idx))


SCRIPT_TABLE = {
    "north_June": _row("sic_pos_sst_neg", [16, 14, 12], [1, 4, 6], _N, standardise=True),      # June1st.py:210-227
    "north_July": _row("pos", [11, 0, 3.125433e+10], [4, 15, 40221.26298973], _N),             # July1st.py:169-179
    "north_August": _row("all_then_pos_p", [9, 7, 3], [4, 13, 13], _N, pthr=0.08),              # August1st.py:169-182
    "north_September": _row("all_then_pos_p", [8, 9, 3], [6, 3, 13], _N, pthr=0.05),            # September1st.py:170-183
    "south_December": _row("pos", [4, 9, 2], [13, 4, 13], _S, drop_first=True),                 # December1st.py:162-171
    "south_January": _row("all_then_pos_p", [2, 1, 3], [14, 14, 14], _S, drop_first=True, pthr=0.08),  # January1st.py:163-175
    "south_February": _row("all_then_pos_p", [16, 5, 3], [0, 11, 13], _S, pthr=0.05),           # February1st.py:162-174
}


def pearson_rp(y, A):
    """Pearson r and two-sided p of y [n] against every row of A [k, n] at once
    (what ``scipy.stats.pearsonr`` returns per area in north/June1st.py:218)."""
    y = np.asarray(y, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64)
    n = y.shape[0]
    ym = y - y.mean()
    Am = A - A.mean(axis=1, keepdims=True)
    r = (Am @ ym) / (np.linalg.norm(Am, axis=1) * np.linalg.norm(ym))
    r = np.clip(r, -1.0, 1.0)
    ab = n / 2.0 - 1.0
    p = 2.0 * special.betainc(ab, ab, 0.5 * (1.0 - np.abs(r)))   # = 2*BetaCDF, what scipy.stats.pearsonr evaluates
    return r, p


def select_features(y, sic_anoms, sst_anoms=None, *, rule, k, pthr=None):
    """Rows of the (n+1)-long area series the reference appends to X for region index k.
    Rules: north/June1st.py:217-224, north/July1st.py:176-179, north/August1st.py:176-182."""
    y0 = np.asarray(y, dtype=np.float64).reshape(-1)
    keys = list(sic_anoms)
    A = np.stack([np.asarray(sic_anoms[a], dtype=np.float64) for a in keys])
    r, p = pearson_rp(y0, A[:, :-1])
    if rule in ("pos", "sic_pos_sst_neg"):
        keep = r > 0
    elif rule == "all_then_pos_p":
        keep = np.ones(len(keys), bool) if k == 0 else (r > 0) & (p / 2 < pthr)
    else:
        raise ValueError("unknown rule %r" % rule)
    feats = [A[i] for i in np.flatnonzero(keep)]
    if rule == "sic_pos_sst_neg":
        keys2 = list(sst_anoms)
        B = np.stack([np.asarray(sst_anoms[a], dtype=np.float64) for a in keys2])
        r2, _ = pearson_rp(y0, B[:, :-1])
        feats += [-B[i] for i in np.flatnonzero(r2 < 0)]
    return feats


def design_matrix(feats, standardise):
    """(X [n,N], Xs [1,N]) from the selected series (north/June1st.py:226-229).  June standardises over
    all n+1 rows including the test row, ddof = 0.  Returned arrays are C-contiguous."""
    if len(feats) == 0:
        raise IndexError("no area passed the feature rule")   # the reference raises IndexError at X[-1,:]
    F = np.asarray(feats, dtype=np.float64).T
    if standardise:
        F = (F - F.mean(0)) / F.std(0)
    return np.ascontiguousarray(F[:-1]), np.ascontiguousarray(F[-1:])


def laplacian_M(X):
    """Negative graph Laplacian of |cov(X)| (north/June1st.py:231-233).

    Uses ``np.cov`` itself, not an algebraically equal formula: for the reference's extreme length scales
    (July region 3, l = 3.1e10) ``expm(l*M)`` amplifies a 1-ulp difference in M to ~1e-6 in
    the forecast, so M has to be bit-identical to the reference's."""
    # the reference's X is a transposed (Fortran-ordered) view; np.cov's BLAS call, and hence its
    # rounding, depends on that layout -- 63/63 golden records are bit-identical in F order, 0/63 in C order
    M = np.atleast_2d(np.abs(np.cov(np.asfortranarray(X, dtype=np.float64), rowvar=False, bias=True)))
    np.fill_diagonal(M, 0.0)
    np.fill_diagonal(M, -np.sum(M, axis=0))
    return M


def sigma_tilde(M, ell):
    """Sigma~ = expm(l M) (north/June1st.py:264).  Kept on SciPy's Pade expm so results match the
    reference for extreme l."""
    return expm(ell * np.asarray(M, dtype=np.float64))


class SigmaEigh:
    """Sigma~(l) = expm(l M) for MANY l on one data set: M is symmetric negative semi-definite,
    so one ``eigh`` per data set gives  Sigma~(l) = Q diag(exp(l lambda)) Q^T  and  M Sigma~(l) = Q diag(lambda
    exp(l lambda)) Q^T  (the derivative d Sigma~/dl the MLII gradient needs) for every l of a grid or an optimiser
    run at the cost of one N x N GEMM each, instead of a Pade ``expm`` per call (twice per fit in the reference,
    north/June1st.py:264, 269).  Agrees with ``scipy.linalg.expm`` to ~1e-13 for moderate l |M|; the single-call
    path keeps ``expm`` so that the reference's own numbers (incl. its l = 3.1e10 table entry) are reproduced."""

    def __init__(self, M):
        M = np.asarray(M, dtype=np.float64)
        self.lam, self.Q = np.linalg.eigh(0.5 * (M + M.T))
        self.lam = np.minimum(self.lam, 0.0)          # the exact spectrum is <= 0; clip rounding above zero

    def sigma(self, ell):
        return (self.Q * np.exp(ell * self.lam)) @ self.Q.T

    def msigma(self, ell):
        return (self.Q * (self.lam * np.exp(ell * self.lam))) @ self.Q.T
