"""The callers of the GP block: the retrospective (region x year) loop and the operational forecast
(north/retrospective_forecasts/September1st_retro.py:171-249, north/June1st.py:208-288), with the
inline GP statements replaced by ``GPR.fit`` / ``GPR.predict``.  Output dicts and ``.round(3)``
semantics are the reference's.
"""
import numpy as np

from .features import LGRID, SGRID, SCRIPT_TABLE, design_matrix, laplacian_M, select_features
from .gpr import GPR
from .smallbatch import SmallBatch


def _problem(tab, k, y, sic, sst):
    """(X, Xs, M) of one (region, year): the statements north/June1st.py:214-233."""
    feats = select_features(y, sic, sst, rule=tab["rule"], k=k, pthr=tab["pthr"])
    X, Xs = design_matrix(feats, tab["standardise"])
    return X, Xs, laplacian_M(X)


def _one(gp, tab, k, y, sic, sst):
    X, Xs, M = _problem(tab, k, y, sic, sst)
    gp.fit(X, y, tab["ell"][k], tab["sn"][k], M=M, Xs=Xs)
    fmean, fvar = gp.predict(Xs)
    return float(fmean[0]), float(fvar[0])


def _retro_inputs(tab, SIC, SIEs_dt, SST, region, year, fmin):
    row = year - (fmin - 1) - 1
    cols = range(1, year - 1979) if tab["drop_first"] else range(year - 1979)               # January1st_retro.py:173
    y = np.asarray([SIEs_dt[region][row, cols]]).T
    key = "anoms_%d" % (year - 1 if tab["drop_first"] else year)
    return row, y, SIC[key], None if SST is None else SST[key]


def retro_forecast(script, SIC, SIEs_dt, SIEs_trend, fmin, fmax, SST=None, gp=None, batched=False):
    """``forecast(fmin, fmax)`` of the retro scripts -> GPR dict of 9 arrays, each (n_years,).

    ``batched=True``: every (region, year) fit of the double loop (September1st_retro.py:176-180) is queued and the lot
    runs in ONE device launch, one workgroup per fit (n = year - 1979 <= 128); ``Sigma~ = expm(l M)`` is still SciPy's
    Pade form per fit, so the numbers are the reference's.  ``batched=False`` is the fit-at-a-time host loop."""
    tab = SCRIPT_TABLE[script]
    own = gp is None
    gp = gp or GPR(kernel="netdiffusion")
    try:
        out = {}
        ny = fmax - fmin + 1
        if batched:
            sb = SmallBatch(gp)
            for k, region in enumerate(tab["regions"]):
                for year in range(fmin, fmax + 1):
                    _, y, sic, sst = _retro_inputs(tab, SIC, SIEs_dt, SST, region, year, fmin)
                    X, Xs, M = _problem(tab, k, y, sic, sst)
                    sb.add_fit(sb.add_dataset(X, y, Xs, M), tab["ell"][k], tab["sn"][k], expm="pade")
            res = sb.run()
            bad = np.flatnonzero(res["info"])
            if len(bad):
                raise np.linalg.LinAlgError("Matrix is not positive definite (fit %d, pivot %d)" % (bad[0], res["info"][bad[0]]))
            for k, region in enumerate(tab["regions"]):
                mu = res["mean"][k * ny:(k + 1) * ny, 0]
                var = res["var"][k * ny:(k + 1) * ny, 0]
                fmean, fvar, fmean_rt = np.round(mu, 3), np.round(var, 3), np.zeros(ny)
                for year in range(fmin, fmax + 1):
                    row, i = year - (fmin - 1) - 1, year - fmin
                    last = (year - 1979) * SIEs_trend[region][row, 0] + SIEs_trend[region][row, 1]
                    fmean_rt[i] = (fmean[i] + last).round(3)
                out[region + "_fmean"], out[region + "_fvar"], out[region + "_fmean_rt"] = fmean, fvar, fmean_rt
            return out
        for k, region in enumerate(tab["regions"]):
            fmean, fvar, fmean_rt = np.zeros(ny), np.zeros(ny), np.zeros(ny)
            for year in range(fmin, fmax + 1):
                row, y, sic, sst = _retro_inputs(tab, SIC, SIEs_dt, SST, region, year, fmin)
                mu, var = _one(gp, tab, k, y, sic, sst)
                i = year - fmin
                fmean[i] = np.round(mu, 3)                                                   # :241
                fvar[i] = np.round(var, 3)                                                   # :242
                lineT = np.arange(year - 1979 + 1) * SIEs_trend[region][row, 0] + SIEs_trend[region][row, 1]
                fmean_rt[i] = (fmean[i] + lineT[-1]).round(3)                                # :244
            out[region + "_fmean"], out[region + "_fvar"], out[region + "_fmean_rt"] = fmean, fvar, fmean_rt
        return out
    finally:
        if own:
            gp.close()


def retro_grid_search(script, SIC, SIEs_dt, fmin, fmax, SST=None, ells=LGRID, sns=SGRID, gp=None):
    """The hyper-parameter search the reference's tables imply (north/June1st.py:210-211: indices into
    ``logspace(-7,2,20) x logspace(-3,9,20)``): nlML (north/June1st.py:246) of every (region, year) of the retro loop at
    every grid point -- 3 x years x 400 fits in one device launch, one eigendecomposition of M per (region, year).
    Returns {region: nlml [n_years, len(ells), len(sns)]} (+inf where K~ is not positive definite)."""
    tab = SCRIPT_TABLE[script]
    own = gp is None
    gp = gp or GPR(kernel="netdiffusion")
    ells = np.asarray(ells, dtype=np.float64).reshape(-1); sns = np.asarray(sns, dtype=np.float64).reshape(-1)
    try:
        sb = SmallBatch(gp)
        ny = fmax - fmin + 1
        for k, region in enumerate(tab["regions"]):
            for year in range(fmin, fmax + 1):
                _, y, sic, sst = _retro_inputs(tab, SIC, SIEs_dt, SST, region, year, fmin)
                X, Xs, M = _problem(tab, k, y, sic, sst)
                ds = sb.add_dataset(X, y, None, M)
                for e in ells:
                    for s_ in sns:
                        sb.add_fit(ds, e, s_, expm="eigh")
        nl = sb.run()["nlml"].reshape(len(tab["regions"]), ny, len(ells), len(sns))
        return {region: nl[k] for k, region in enumerate(tab["regions"])}
    finally:
        if own:
            gp.close()


def retro_optimise(script, SIC, SIEs_dt, fmin, fmax, SST=None, SIEs_trend=None, gp=None, theta0=None, expm="eigh", **kw):
    """The optimiser call the reference left commented out (north/June1st.py:259-262, `minimize(MLII, x0, method='CG', jac=True)`
    with x0 = log of the script's table entries), re-enabled for EVERY (region, year) of the retro loop
    (September1st_retro.py:176-180) at once: each optimiser round is one device launch, one workgroup per (region, year), with
    nlML (:246) and its exact gradient formed in LDS (``GPR.optimize_batch``; the reference's own "gradient" formulae, :248-252,
    are not a derivative -- SURVEY App. C-7 -- and are available through ``nlml_batch(grad='ref')``).
    Returns {region: dict(x [n_years, 2] = (log l, log sn~), fun, nit, converged)} and ``nfev`` (device launches); with
    ``SIEs_trend`` also the forecast at each optimum in the reference's GPR-dict layout (``_fmean``, ``_fvar``, ``_fmean_rt``,
    rounded as :241-244)."""
    tab = SCRIPT_TABLE[script]
    own = gp is None
    gp = gp or GPR(kernel="netdiffusion")
    try:
        ny = fmax - fmin + 1
        Xl, yl, Xsl, Ml, x0 = [], [], [], [], []
        for k, region in enumerate(tab["regions"]):
            for year in range(fmin, fmax + 1):
                _, y, sic, sst = _retro_inputs(tab, SIC, SIEs_dt, SST, region, year, fmin)
                X, Xs, M = _problem(tab, k, y, sic, sst)
                Xl.append(X); yl.append(y); Xsl.append(Xs); Ml.append(M)
                x0.append([np.log(tab["ell"][k]), np.log(tab["sn"][k])])
        x0 = np.asarray(x0) if theta0 is None else np.broadcast_to(np.asarray(theta0, dtype=np.float64), (len(Xl), 2))
        res = gp.optimize_batch(Xl, yl, x0, M=Ml, expm=expm, **kw)
        out = {"nfev": res["nfev"]}
        for k, region in enumerate(tab["regions"]):
            sl = slice(k * ny, (k + 1) * ny)
            out[region] = dict(x=res["x"][sl], fun=res["fun"][sl], nit=res["nit"][sl], converged=res["converged"][sl])
        if SIEs_trend is not None:
            th = res["x"]
            r = gp.fit_batch(Xl, yl, Xsl, np.exp(th[:, 0]), np.exp(th[:, 1]), M=Ml)
            for k, region in enumerate(tab["regions"]):
                fmean, fvar = np.round(r["mean"][k * ny:(k + 1) * ny, 0], 3), np.round(r["var"][k * ny:(k + 1) * ny, 0], 3)
                fmean_rt = np.zeros(ny)
                for year in range(fmin, fmax + 1):
                    row, i = year - (fmin - 1) - 1, year - fmin
                    fmean_rt[i] = (fmean[i] + (year - 1979) * SIEs_trend[region][row, 0] + SIEs_trend[region][row, 1]).round(3)
                out[region + "_fmean"], out[region + "_fvar"], out[region + "_fmean_rt"] = fmean, fvar, fmean_rt
        return out
    finally:
        if own:
            gp.close()


def operational_forecast(script, SIC, SIEs_dt, SIEs_trend, ymax, SST=None, gp=None):
    """``forecast(ymax)`` of the operational scripts -> {region: dict(fmean, fvar, fmean_rt)} (unrounded)."""
    tab = SCRIPT_TABLE[script]
    own = gp is None
    gp = gp or GPR(kernel="netdiffusion")
    try:
        out = {}
        for k, region in enumerate(tab["regions"]):
            yv = SIEs_dt[region][1:] if tab["drop_first"] else SIEs_dt[region]              # December1st.py:165
            y = np.asarray([yv]).T
            mu, var = _one(gp, tab, k, y, SIC["anoms"], None if SST is None else SST["anoms"])
            lineT = np.arange(ymax - 1979 + 1) * SIEs_trend[region][0] + SIEs_trend[region][1]
            out[region] = dict(fmean=mu, fvar=var, fmean_rt=mu + lineT[-1])
        return out
    finally:
        if own:
            gp.close()
