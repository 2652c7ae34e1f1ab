"""The callers of the GP block: the retrospective (region x year) loop and the operational forecast
(north/retrospective_forecasts/September1st_retro.py:171-249, north/June1st.py:208-288), with the
inline GP statements replaced by ``GPR.fit`` / ``GPR.predict``.  Output dicts and ``.round(3)``
semantics are the reference's.
"""
import numpy as np

from .features import SCRIPT_TABLE, design_matrix, laplacian_M, select_features
from .gpr import GPR


def _one(gp, tab, k, y, sic, sst):
    feats = select_features(y, sic, sst, rule=tab["rule"], k=k, pthr=tab["pthr"])
    X, Xs = design_matrix(feats, tab["standardise"])
    M = laplacian_M(X)
    gp.fit(X, y, tab["ell"][k], tab["sn"][k], M=M, Xs=Xs)
    fmean, fvar = gp.predict(Xs)
    return float(fmean[0]), float(fvar[0])


def retro_forecast(script, SIC, SIEs_dt, SIEs_trend, fmin, fmax, SST=None, gp=None):
    """``forecast(fmin, fmax)`` of the retro scripts -> GPR dict of 9 arrays, each (n_years,)."""
    tab = SCRIPT_TABLE[script]
    own = gp is None
    gp = gp or GPR(kernel="netdiffusion")
    try:
        out = {}
        ny = fmax - fmin + 1
        for k, region in enumerate(tab["regions"]):
            fmean, fvar, fmean_rt = np.zeros(ny), np.zeros(ny), np.zeros(ny)
            for year in range(fmin, fmax + 1):
                row = year - (fmin - 1) - 1
                cols = range(1, year - 1979) if tab["drop_first"] else range(year - 1979)   # January1st_retro.py:173
                y = np.asarray([SIEs_dt[region][row, cols]]).T
                key = "anoms_%d" % (year - 1 if tab["drop_first"] else year)
                mu, var = _one(gp, tab, k, y, SIC[key], None if SST is None else SST[key])
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
