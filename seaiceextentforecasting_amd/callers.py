"""The callers immediately before and after the GP block, restated with the reference's
outputs: per-pixel linear detrending (the step before the ComplexNetworks feature pipeline) and the retro
``skill()`` scores + the two CSV tables.  Host-side NumPy: these are O(pixels x years) / O(years) and feed or
consume the HIP engine; parity is pinned by tests/golden/callers.npz (the reference's own functions run on
synthetic inputs)."""
import numpy as np


def detrend_cube(data):
    """Per-pixel least-squares line removal over the last axis, all pixels at once.

    Restates ``detrend`` (north/June1st.py:179-194): pixels that are all-NaN stay NaN; a pixel with some NaN
    gets NaN slope/intercept (``scipy.stats.linregress`` propagates NaN), hence an all-NaN detrended series.
    Returns (detrended [X,Y,T], trend [X,Y,2] = slope, intercept)."""
    data = np.asarray(data, dtype=np.float64)
    X, Y, T = data.shape
    t = np.arange(T, dtype=np.float64)
    tm = t.mean()
    ssxm = np.mean((t - tm) ** 2)
    ym = data.mean(axis=2)                                   # NaN wherever a pixel has any NaN
    ssxym = np.mean((t - tm) * (data - ym[:, :, None]), axis=2)
    slope = ssxym / ssxm
    intercept = ym - slope * tm
    allnan = np.isnan(data).all(axis=2)
    slope = np.where(allnan, np.nan, slope)
    intercept = np.where(allnan, np.nan, intercept)
    detrended = data - (slope[:, :, None] * t + intercept[:, :, None])
    return detrended, np.stack([slope, intercept], axis=2)


def detrend(dataset, fmin=None, fmax=None, engine=None):
    """In-place drop-in for the reference's ``detrend``: operational form ``detrend(dataset)`` sets
    ``dataset['dt'], dataset['trend']`` (north/June1st.py:179-194); retro form ``detrend(dataset, fmin, fmax)``
    sets ``dataset['dt_YYYY'], dataset['trend_YYYY']`` from the first YYYY-1979+1 years
    (north/retrospective_forecasts/June1st_retro.py:178-195).  ``engine``: a ``GPR`` handle -- every cut-off year is then
    detrended on the GPU in one launch (``sigp_detrend``)."""
    if fmin is None:
        if engine is not None:
            (dt,), (tr,) = engine.detrend_cuts(dataset["data"], [np.asarray(dataset["data"]).shape[2]])
            dataset["dt"], dataset["trend"] = dt, tr
        else:
            dataset["dt"], dataset["trend"] = detrend_cube(dataset["data"])
        return dataset
    years = list(range(fmin, fmax + 1))
    if engine is not None:
        dts, trs = engine.detrend_cuts(dataset["data"], [year - 1979 + 1 for year in years])
        for year, dt, tr in zip(years, dts, trs):
            dataset["dt_%d" % year], dataset["trend_%d" % year] = dt, tr
        return dataset
    for year in years:
        n = year - 1979 + 1
        dataset["dt_%d" % year], dataset["trend_%d" % year] = detrend_cube(dataset["data"][:, :, :n])
    return dataset


def _mse_skill(obs, forecast):
    """1 - MSE(forecast) / MSE(climatology), climatology = the mean of the observations."""
    obs = np.asarray(obs, dtype=np.float64)
    err = np.mean((obs - forecast) ** 2)
    clim = np.mean((obs - np.nanmean(obs)) ** 2)
    return (1 - (err / clim)).round(3)


def skill(GPR, SIEs, SIEs_dt, fmin, fmax, regions):
    """MSE skill scores of the retrended and detrended forecasts, rounded to 3 decimals (behaviour of
    north/retrospective_forecasts/June1st_retro.py:293-314).  The detrended observation of forecast year t is the
    entry of year t in the series detrended with cut-off t.  Returns (skill_rt, skill_dt, dt_obs)."""
    years = np.arange(fmin, fmax + 1)
    skill_rt, skill_dt, dt_obs = [], [], []
    for region in regions:
        dt = list(np.asarray(SIEs_dt[region])[years - (fmin - 1), years - 1979])
        dt_obs.append(dt)
        skill_rt.append(_mse_skill(np.asarray(SIEs[region])[fmin - 1979:], GPR[region + "_fmean_rt"]))
        skill_dt.append(_mse_skill(dt, GPR[region + "_fmean"]))
    return skill_rt, skill_dt, dt_obs


def forecast_tables(GPR, SIEs, SIEs_dt, fmin, fmax, regions):
    """The two DataFrames the retro scripts write as CSV (June1st_retro.py:346-369): detrended forecasts with
    uncertainty (sqrt(fvar).round(3)) and forecasts with trend, one row per year plus a final 'Skill' row."""
    import pandas as pd
    skill_rt, skill_dt, dt_obs = skill(GPR, SIEs, SIEs_dt, fmin, fmax, regions)
    years = list(range(fmin, fmax + 1)) + ["Skill"]

    def prep(data, sk=None):
        data = list(np.asarray(data).tolist())
        data.append("" if sk is None else sk)
        return data

    cols_dt, cols_rt, data_dt, data_rt = [], [], [], []
    for k, r in enumerate(regions):
        cols_dt += [r + "$_o$", r + "$_f$", r + "$_f$ unc"]
        data_dt += [prep(dt_obs[k]), prep(GPR[r + "_fmean"], skill_dt[k]), prep(np.sqrt(GPR[r + "_fvar"]).round(3))]
        cols_rt += [r + "$_o$", r + "$_f$"]
        data_rt += [prep(np.asarray(SIEs[r])[fmin - 1979:]), prep(GPR[r + "_fmean_rt"], skill_rt[k])]
    df_dt = pd.DataFrame(list(zip(*data_dt)), index=years, columns=cols_dt)
    df_rt = pd.DataFrame(list(zip(*data_rt)), index=years, columns=cols_rt)
    return df_dt, df_rt
