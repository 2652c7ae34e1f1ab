#!/usr/bin/env python3
"""Goldens for the callers either side of the GP block: the reference's own ``skill()`` and
``detrend()`` FunctionDefs, AST-extracted from the scripts under /root/reference and executed on synthetic inputs
(authoring container only; numeric inputs/outputs only are written)."""
import ast
import os

import numpy as np
from scipy.stats import linregress

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    src = open(os.path.join(REF, path)).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    return compile(ast.Module([fn], []), "<ref %s %s>" % (name, path), "exec")


def main():
    rng = np.random.default_rng(99)
    out = {}
    # ---- skill(): north June retro and south February retro ------------------------------------------------
    for tag, path, regions in (("north", "north/retrospective_forecasts/June1st_retro.py", ["Pan-Arctic", "Beaufort", "Chukchi"]),
                               ("south", "south/retrospective_forecasts/February1st_retro.py", ["Pan-Antarctic", "Ross", "Weddell"])):
        fmin, fmax = 1995, 2003
        ny, T = fmax - fmin + 1, fmax - 1979 + 1
        SIEs = {r: 7.0 - 0.05 * np.arange(T) + 0.4 * rng.standard_normal(T) for r in regions}
        SIEs_dt = {r: 0.4 * rng.standard_normal((fmax - fmin + 2, T)) for r in regions}
        GPR = {}
        for r in regions:
            GPR[r + "_fmean"] = np.round(0.3 * rng.standard_normal(ny), 3)
            GPR[r + "_fvar"] = np.round(0.05 + 0.02 * rng.random(ny), 3)
            GPR[r + "_fmean_rt"] = np.round(SIEs[r][fmin - 1979:] + 0.2 * rng.standard_normal(ny), 3)
        ns = dict(np=np, SIEs=SIEs, SIEs_dt=SIEs_dt, GPR=GPR)
        exec(load(path, "skill"), ns)
        skill_rt, skill_dt, dt_obs = ns["skill"](fmin, fmax)
        out["skill_%s/args" % tag] = np.array([fmin, fmax])
        for r in regions:
            out["skill_%s/SIEs/%s" % (tag, r)] = SIEs[r]
            out["skill_%s/SIEs_dt/%s" % (tag, r)] = SIEs_dt[r]
            for k in ("_fmean", "_fvar", "_fmean_rt"):
                out["skill_%s/GPR/%s%s" % (tag, r, k)] = GPR[r + k]
        out["skill_%s/skill_rt" % tag] = np.array(skill_rt, dtype=np.float64)
        out["skill_%s/skill_dt" % tag] = np.array(skill_dt, dtype=np.float64)
        out["skill_%s/dt_obs" % tag] = np.array(dt_obs, dtype=np.float64)
    # ---- detrend(): retro (one detrended cube per cut-off year) and operational ----------------------------
    X, Y, T = 7, 6, 30
    data = rng.standard_normal((X, Y, T)) + 0.03 * np.arange(T)
    data[0, 0, :] = np.nan                       # land / no-ice pixel: stays NaN
    data[3, 2, 5] = np.nan                       # a pixel with one missing year: linregress propagates NaN
    ds = {"data": data.copy()}
    ns = dict(np=np, linregress=linregress)
    exec(load("north/retrospective_forecasts/June1st_retro.py", "detrend"), ns)
    fmin, fmax = 1979 + 25, 1979 + 29
    ns["detrend"](ds, fmin, fmax)
    out["detrend_retro/data"] = data
    out["detrend_retro/args"] = np.array([fmin, fmax])
    for year in range(fmin, fmax + 1):
        out["detrend_retro/dt_%d" % year] = ds["dt_%d" % year]
        out["detrend_retro/trend_%d" % year] = ds["trend_%d" % year]
    ds2 = {"data": data.copy()}
    ns2 = dict(np=np, linregress=linregress)
    exec(load("north/June1st.py", "detrend"), ns2)
    ns2["detrend"](ds2)
    out["detrend_op/dt"] = ds2["dt"]
    out["detrend_op/trend"] = ds2["trend"]
    np.savez_compressed(os.path.join(OUT, "callers.npz"), **out)
    print("skill north", out["skill_north/skill_rt"], out["skill_north/skill_dt"], "; detrend keys", [k for k in out if k.startswith("detrend")][:4])


if __name__ == "__main__":
    main()
