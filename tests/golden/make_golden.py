#!/usr/bin/env python3
"""Generate golden vectors for the GPR hot path by running the REFERENCE's own ``forecast()``.

Runs only in the authoring container (it reads /root/reference by path; the reference never
travels).  For each of the 14 forecast scripts it parses the file with ``ast``, extracts the
``FunctionDef forecast`` node, and executes it -- unmodified -- in a namespace seeded with NumPy/SciPy
and *synthetic* ``SIC / SST / SIEs_dt / SIEs_trend`` globals.  A line tracer
snapshots the frame's full-precision locals for every (region k, year) iteration and calls the live
``MLII`` closure at several theta (including one that takes the ``inf`` error branch).

Only numeric inputs/outputs are written (``tests/golden/<script>.npz``); no reference source or
bytecode is copied.  Library versions at generation time are recorded in each file.

    python tests/golden/make_golden.py            # regenerates every fixture
"""
import ast
import os
import sys

import numpy as np
import scipy
from scipy.linalg import expm
from scipy.optimize import minimize
from scipy.stats import pearsonr

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# script key -> (path, kind, uses_sst, south_prev_year)
SCRIPTS = {
    "north_June": ("north/June1st.py", "op", True, False),
    "north_July": ("north/July1st.py", "op", False, False),
    "north_August": ("north/August1st.py", "op", False, False),
    "north_September": ("north/September1st.py", "op", False, False),
    "south_December": ("south/December1st.py", "op", False, True),
    "south_January": ("south/January1st.py", "op", False, True),
    "south_February": ("south/February1st.py", "op", False, False),
    "north_June_retro": ("north/retrospective_forecasts/June1st_retro.py", "retro", True, False),
    "north_July_retro": ("north/retrospective_forecasts/July1st_retro.py", "retro", False, False),
    "north_August_retro": ("north/retrospective_forecasts/August1st_retro.py", "retro", False, False),
    "north_September_retro": ("north/retrospective_forecasts/September1st_retro.py", "retro", False, False),
    "south_December_retro": ("south/retrospective_forecasts/December1st_retro.py", "retro", False, True),
    "south_January_retro": ("south/retrospective_forecasts/January1st_retro.py", "retro", False, True),
    "south_February_retro": ("south/retrospective_forecasts/February1st_retro.py", "retro", False, False),
}
NORTH = ["Pan-Arctic", "Beaufort", "Chukchi"]
SOUTH = ["Pan-Antarctic", "Ross", "Weddell"]

CAPTURE = ["X", "Xs", "y", "M", "Σ_tilde", "L_tilde", "A_tilde", "σf", "σn", "Σ", "L", "α",
           "KXXs", "KXsXs", "v", "l", "σn_tilde"]
ASCII = {"Σ_tilde": "Sigma_tilde", "σf": "sigma_f", "σn": "sigma_n", "Σ": "Sigma", "α": "alpha",
         "l": "ell", "σn_tilde": "sn_tilde"}


def load_forecast(path):
    src = open(os.path.join(REF, path)).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "forecast")
    return compile(ast.Module([fn], []), "<ref forecast %s>" % path, "exec")


def make_anoms(rng, signals, length, n_areas, sign=+1.0, off=0):
    """Area anomaly series correlated with the regional signals (so the r>0 & p/2<thr rules select)."""
    d = {}
    for a in range(n_areas):
        s = signals[a % len(signals)][off:off + length]
        c = 0.5 + rng.random()
        d[a * 3 + 1] = sign * c * s + 0.45 * rng.standard_normal(length)   # area ids: arbitrary ints
    return d


def synth_inputs(kind, uses_sst, prev_year, south, seed, n_areas, fmin, fmax, ymax):
    rng = np.random.default_rng(seed)
    regions = SOUTH if south else NORTH
    T = (fmax if kind == "retro" else ymax) - 1979 + 2
    base = np.cumsum(rng.standard_normal(T)) * 0.3 + rng.standard_normal(T)
    signals = [base, 0.7 * base + 0.5 * rng.standard_normal(T), 0.6 * base + 0.6 * rng.standard_normal(T)]
    SIC, SST, SIEs_dt, SIEs_trend = {}, ({} if uses_sst else None), {}, {}
    if kind == "op":
        # operational: y = SIEs_dt[region] (len n) or [1:] for south Dec/Jan; anoms have len(y)+1 entries
        ylen = ymax - 1979                       # north: n = ymax-1979 targets
        for k, r in enumerate(regions):
            SIEs_dt[r] = signals[k][:ylen].copy()
            SIEs_trend[r] = np.array([-0.05 - 0.01 * k, 7.0 - k])
        alen = (ylen - 1 if prev_year else ylen) + 1
        off = 1 if prev_year else 0              # south Dec/Jan: y drops the first year (December1st.py:165)
        SIC["anoms"] = make_anoms(rng, signals, alen, n_areas, off=off)
        if uses_sst:
            SST["anoms"] = make_anoms(rng, signals, alen, max(3, n_areas // 2), sign=-1.0)
    else:
        rows = fmax - fmin + 2
        for k, r in enumerate(regions):
            SIEs_dt[r] = np.stack([signals[k] + 0.05 * i * rng.standard_normal(T) for i in range(rows)])
            SIEs_trend[r] = np.stack([np.array([-0.05 - 0.01 * k, 7.0 - k]) + 0.001 * i for i in range(rows)])
        for year in range(fmin - 1, fmax + 1):
            SIC["anoms_%d" % year] = make_anoms(rng, signals, year - 1979 + 1, n_areas, off=1 if prev_year else 0)
            if uses_sst:
                SST["anoms_%d" % year] = make_anoms(rng, signals, year - 1979 + 1, max(3, n_areas // 2), sign=-1.0)
    return SIC, SST, SIEs_dt, SIEs_trend


def run_reference(code, kind, args, glb, thetas_extra):
    ns = dict(np=np, expm=expm, pearsonr=pearsonr, minimize=minimize, print=lambda *a, **k: None)
    ns.update(glb)
    exec(code, ns)
    records = []
    state = {"last_v": None}

    def local(frame, event, arg):
        loc = frame.f_locals
        v = loc.get("v")
        if v is not None and v is not state["last_v"]:
            state["last_v"] = v
            rec = {"k": int(loc["k"]), "year": int(loc["year"]) if "year" in loc else -1}
            for name in CAPTURE:
                rec[ASCII.get(name, name)] = np.array(loc[name], dtype=np.float64, copy=True)
            th0 = np.array([np.log(float(loc["l"])), np.log(float(loc["σn_tilde"]))])
            thetas = [th0, th0 + np.array([0.3, -0.2]), th0 + np.array([-1.0, 0.7]),
                      np.array([np.log(1.8e-3), np.log(0.33)]), np.array([np.log(0.14), np.log(6.1)])]
            thetas += thetas_extra
            sys.settrace(None)
            try:
                vals = [loc["MLII"](t) for t in thetas]
            finally:
                sys.settrace(tracer)
            rec["mlii_theta"] = np.array(thetas)
            rec["mlii_nlml"] = np.array([float(a) for a, _ in vals])
            rec["mlii_grad"] = np.array([np.asarray(g, dtype=np.float64) for _, g in vals])
            rec["_frame_done"] = False
            records.append(rec)
        if records:
            rec = records[-1]
            try:
                if kind == "op":
                    if int(loc["k"]) == rec["k"] and "fmean_rt" in loc:
                        rec["fmean"] = float(loc["fmean"]); rec["fvar"] = float(loc["fvar"])
                        rec["fmean_rt"] = float(loc["fmean_rt"])
                else:
                    i = rec["year"] - args[0]
                    if int(loc["k"]) == rec["k"]:
                        rec["fmean_r3"] = float(loc["fmean"][i]); rec["fvar_r3"] = float(loc["fvar"][i])
                        rec["fmean_rt_r3"] = float(loc["fmean_rt"][i])
            except (KeyError, TypeError, IndexError):
                pass
        return local

    def tracer(frame, event, arg):
        if event == "call" and frame.f_code.co_name == "forecast":
            return local
        return None

    sys.settrace(tracer)
    try:
        ret = ns["forecast"](*args)
    finally:
        sys.settrace(None)
    return ret, records


def pack_anoms(prefix, dd, out):
    for key, d in dd.items():
        ids = list(d.keys())
        out["%s/%s/ids" % (prefix, key)] = np.array(ids, dtype=np.int64)
        out["%s/%s/data" % (prefix, key)] = np.stack([d[i] for i in ids])


def main():
    os.chdir("/tmp")
    for idx, (name, (path, kind, uses_sst, prev_year)) in enumerate(SCRIPTS.items()):
        south = name.startswith("south")
        code = load_forecast(path)
        seed = 7100 + idx
        n_areas = [9, 12, 14, 10, 8, 11, 13][idx % 7]
        if kind == "op":
            ymax = 2013 + (idx % 5)
            fmin = fmax = None
            args = (ymax,)
        else:
            fmin, fmax = 2003 + (idx % 3), 2004 + (idx % 3)
            ymax = None
            args = (fmin, fmax)
        SIC, SST, SIEs_dt, SIEs_trend = synth_inputs(kind, uses_sst, prev_year, south, seed, n_areas, fmin, fmax, ymax)
        glb = dict(SIC=SIC, SIEs_dt=SIEs_dt, SIEs_trend=SIEs_trend)
        if uses_sst:
            glb["SST"] = SST
        fail_theta = [np.array([np.log(1e-3), -80.0])]     # sn~ -> 0 with rank-deficient X Sigma X^T: inf branch
        ret, records = run_reference(code, kind, args, glb, fail_theta)
        out = {"meta/script": np.array(name), "meta/kind": np.array(kind),
               "meta/args": np.array(args, dtype=np.int64),
               "meta/versions": np.array("numpy %s scipy %s python %s" % (np.__version__, scipy.__version__,
                                                                            sys.version.split()[0]))}
        pack_anoms("SIC", SIC, out)
        if uses_sst:
            pack_anoms("SST", SST, out)
        for r in SIEs_dt:
            out["SIEs_dt/" + r] = np.asarray(SIEs_dt[r], dtype=np.float64)
            out["SIEs_trend/" + r] = np.asarray(SIEs_trend[r], dtype=np.float64)
        out["meta/nrec"] = np.array(len(records))
        for j, rec in enumerate(records):
            for key, val in rec.items():
                if key.startswith("_"):
                    continue
                out["rec%02d/%s" % (j, key)] = np.asarray(val)
        if kind == "retro":
            for key, val in ret.items():
                out["GPR/" + key] = np.asarray(val)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        nfail = sum(int(np.isinf(r["mlii_nlml"]).sum()) for r in records)
        print("%-24s %-5s records=%2d  n=%s  N=%s  inf-branch hits=%d" % (
            name, kind, len(records), sorted({r["X"].shape[0] for r in records}),
            sorted({r["X"].shape[1] for r in records}), nfail))


if __name__ == "__main__":
    main()
