"""GPU tier, round 5: the reference's MLII closure (north/June1st.py:235-257) batched for the reference's OWN kernel -- value,
the reference's "gradient" formulae and the exact derivative for every (data set, theta) of a list in ONE launch -- the lockstep
optimiser over the retro loop's (region, year) data sets (the call the reference left commented out, :259-262), and the bench's
own step shapes against the oracle.

Tolerances as in test_hip_parity.py (predictions <= 1e-8 relative, nlML / sigma_f <= 1e-9) unless stated.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_NAMES, load_golden
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def S():
    import seaiceextentforecasting_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def all_records():
    recs = []
    for name in GOLDEN_NAMES:
        for r in load_golden(name)["records"]:
            recs.append((name, r))
    return recs


# ---- a9 batched: the reference's live closure, every golden record x every theta, ONE launch ---------------------------------
@pytest.mark.parametrize("expm", ["eigh", "pade"])
def test_mlii_closure_all_golden_records_in_one_launch(S, all_records, expm):
    """63 (region, year) records x 6 theta of the reference's live ``MLII`` closure (captured by tests/golden/make_golden.py) through
    ``sigp_small_run_grad``: one workgroup per (record, theta), one launch.  Value <= 1e-9; the reference's "gradient" (:248-252) to the
    tolerance of test_mlii_contract_against_reference_closure (both sides carry cond(K~) eps: the reference sums
    tr(solve(L.T, solve(L, dK))) in LU arithmetic); the except branch -> (inf, [inf, inf]) (:254-256)."""
    assert len(all_records) == 63
    with S.GPR(kernel="netdiffusion") as gp:
        sb = S.SmallBatch(gp)
        want = []
        for _, r in all_records:
            ds = sb.add_dataset(r["X"], r["y"], r["Xs"], r["M"])
            for th, nl, gr in zip(r["mlii_theta"], r["mlii_nlml"], r["mlii_grad"]):
                ell, sn = float(np.exp(th[0])), float(np.exp(th[1]))
                if expm == "eigh" and ell > 1e6:
                    continue          # l = 3.1e10: the eigen route and scipy's Pade expm differ beyond ~1e-10 there
                sb.add_fit(ds, ell, sn, expm=expm)
                want.append((r, th, nl, gr))
        assert len(want) >= 63 * 5
        gp.profile(True, ["small"])
        launches0 = gp.profile_get()["small"]["launches"]
        res = sb.run(grad=True)
        assert gp.profile_get()["small"]["launches"] - launches0 == 1          # ONE launch for the lot
    ninf = 0
    for i, (r, th, nl, gr) in enumerate(want):
        if np.isinf(nl):
            ninf += 1
            assert res["info"][i] > 0 and np.isinf(res["nlml"][i]) and np.all(np.isinf(res["grad_ref"][i])) and np.all(np.isinf(res["grad_exact"][i])), (i, th)
            continue
        assert res["info"][i] == 0
        if np.exp(th[0]) > 1e6:
            assert np.all(np.isfinite(res["grad_ref"][i]))
            continue          # expm(l M) moves by ~1e-6 under 1-ulp changes of l there: as in the single-fit test
        assert abs(res["nlml"][i] - nl) <= 1e-9 * abs(nl), (i, th, res["nlml"][i], nl)
        Kt = O.fit_predict(r["X"], r["y"], r["Xs"], float(np.exp(th[0])), float(np.exp(th[1])), M=r["M"], ref_idiom=False)["K_tilde"]
        tol = max(1e-7, 1e3 * 2.3e-16 * np.linalg.cond(Kt))
        assert np.max(np.abs(res["grad_ref"][i] - gr)) <= tol * max(1.0, np.max(np.abs(gr))), (i, th, res["grad_ref"][i], gr)
    assert ninf == 63           # every record carries one theta of the except branch


def test_batched_exact_gradient_matches_oracle_and_single_fit_path(S, all_records):
    """grad_exact of the batched kernel == O.mlii(grad='exact') == the blocked engine's single-fit gp.nlml(grad='exact'), on golden
    records and on ragged synthetic sets up to n = 128 (N > n and N < n, ch-chunked features)."""
    rng = np.random.default_rng(21)
    sets = [(r["X"], r["y"][:, 0], r["M"]) for _, r in all_records[::9]]
    for n, N in ((128, 60), (100, 130), (45, 200), (2, 1), (7, 3), (64, 33)):
        X = rng.standard_normal((n, N)); y = X @ rng.standard_normal(N) / np.sqrt(N) + 0.3 * rng.standard_normal(n)
        sets.append((X, y, None))
    thetas = [np.array([np.log(0.14), np.log(6.1)]), np.array([np.log(1.8e-3), np.log(0.33)]), np.array([np.log(0.02), np.log(0.05)])]
    with S.GPR(kernel="netdiffusion") as gp:
        gp.upload_batch([s[0] for s in sets], [s[1] for s in sets], None, M=[s[2] for s in sets])
        B = len(sets)
        th = np.array([t for t in thetas for _ in range(B)])                     # pair i -> data set i % B
        val, g = gp.nlml_batch(th, grad="exact")
        val_r, g_r = gp.nlml_batch(th, grad="ref")
        val_p, g_p = gp.nlml_batch(th, grad="exact", expm="pade")
        assert np.array_equal(val, val_r)
    for i in range(len(th)):
        X, y, M = sets[i % B]
        fo, go = O.mlii(th[i], X, y, M=M, grad="exact")
        fr, gr = O.mlii(th[i], X, y, M=M, grad="ref")
        Kt = O.fit_predict(X, y, np.zeros((1, X.shape[1])), float(np.exp(th[i][0])), float(np.exp(th[i][1])), M=M, ref_idiom=False)["K_tilde"]
        tol = max(1e-7, 1e3 * 2.3e-16 * np.linalg.cond(Kt))
        assert abs(val[i] - fo) <= 1e-9 * abs(fo), (i, X.shape)
        assert np.max(np.abs(g[i] - go)) <= tol * max(1.0, np.max(np.abs(go))), (i, X.shape, g[i], go)
        assert np.max(np.abs(g_r[i] - gr)) <= tol * max(1.0, np.max(np.abs(gr))), (i, X.shape, g_r[i], gr)
        assert abs(val_p[i] - fo) <= 1e-9 * abs(fo) and np.max(np.abs(g_p[i] - go)) <= tol * max(1.0, np.max(np.abs(go))), (i, X.shape, g_p[i], go)
    X, y, M = sets[0]
    with S.GPR(kernel="netdiffusion") as gp:                                   # the single-fit path of the blocked engine
        gp.set_data(X, y, M=M)
        f1, g1 = gp.nlml(thetas[0], grad="exact")
    assert abs(f1 - val[0]) <= 1e-9 * abs(f1) and np.allclose(g1, g[0], rtol=1e-7, atol=1e-8)
    # finite differences of the batched value itself
    with S.GPR(kernel="netdiffusion") as gp:
        gp.upload_batch([X], [y], None, M=[M])
        h = 1e-5
        pts = np.array([thetas[0] + h * e for e in np.eye(2)] + [thetas[0] - h * e for e in np.eye(2)])
        v, _ = gp.nlml_batch(pts, grad=None)
        fd = (v[:2] - v[2:]) / (2 * h)
    assert np.allclose(g[0], fd, rtol=1e-5, atol=1e-6), (g[0], fd)


def test_batched_mlii_overflow_and_bad_arguments(S):
    rng = np.random.default_rng(4)
    X = rng.standard_normal((30, 5)); y = rng.standard_normal(30)
    with S.GPR(kernel="netdiffusion") as gp:
        with pytest.raises(RuntimeError):
            gp.nlml_batch(np.zeros((1, 2)))
        gp.upload_batch([X], [y], None)
        v, g = gp.nlml_batch(np.array([[800.0, 0.0], [0.0, 800.0], [np.log(0.1), 0.0]]), grad="ref")     # exp overflows: :254-256
        assert np.isinf(v[0]) and np.isinf(v[1]) and np.all(np.isinf(g[:2])) and np.isfinite(v[2]) and np.all(np.isfinite(g[2]))
        with pytest.raises(ValueError):
            gp.nlml_batch(np.zeros((1, 2)), grad="both")
        with pytest.raises(ValueError):
            gp.nlml_batch(np.zeros((1, 3)))
    with S.GPR(kernel="rbf") as gp:
        with pytest.raises(ValueError):
            gp.nlml_batch(np.zeros((1, 2)), grad="ref")


# ---- f1: the reference's optimiser call for its own kernel, every (region, year) of a retro run in lockstep ---------------------
@pytest.mark.parametrize("script", ["north_September", "south_January", "north_June"])
def test_retro_optimiser_lockstep_reaches_the_single_fit_stationary_points(S, script):
    """``retro_optimise``: the 3 regions x years optimisers of a retro run (September1st_retro.py:176-180 with :259-262 re-enabled)
    advance together, one launch per round; <= 40 device calls; every data set ends at a stationary point of ITS nlML (checked with the
    oracle's exact gradient), at or below the value scipy's L-BFGS-B reaches from the same x0 on the single-fit path of the blocked
    engine (``GPR.optimize``), and the same driver run ONE data set at a time through ``GPR.nlml`` lands on the same point."""
    from seaiceextentforecasting_amd.optim import newton_lockstep
    from seaiceextentforecasting_amd.retro import _problem, _retro_inputs
    g = load_golden(script + "_retro")
    fmin, fmax = g["args"]
    out = S.retro_optimise(script, g["SIC"], g["SIEs_dt"], fmin, fmax, SST=g["SST"], SIEs_trend=g["SIEs_trend"], maxiter=40)
    assert out["nfev"] <= 40, out["nfev"]
    tab = S.SCRIPT_TABLE[script]
    checked = 0
    for k, region in enumerate(tab["regions"]):
        r = out[region]
        assert np.all(r["converged"]), (region, r)
        assert out[region + "_fmean"].shape == (fmax - fmin + 1,) and np.all(np.isfinite(out[region + "_fvar"]))
        for year in range(fmin, fmax + 1):
            i = year - fmin
            _, y, sic, sst = _retro_inputs(tab, g["SIC"], g["SIEs_dt"], g["SST"], region, year, fmin)
            X, Xs, M = _problem(tab, k, y, sic, sst)
            fo, go = O.mlii(r["x"][i], X, y, M=M, grad="exact")
            assert abs(fo - r["fun"][i]) <= 1e-8 * max(1.0, abs(fo))
            assert np.max(np.abs(go)) <= 1e-3 * max(1.0, abs(fo)), (region, year, go)         # stationary for the oracle too
            x0 = np.array([np.log(tab["ell"][k]), np.log(tab["sn"][k])])
            with S.GPR(kernel="netdiffusion", expm="eigh") as gp:
                gp.set_data(X, y, M=M)
                single = gp.optimize(x0)                                                       # scipy L-BFGS-B, blocked engine
                # both stop at |g| <= 1e-5; along the flat l -> 0 valleys of these likelihoods that leaves f within ~gtol of its limit
                assert r["fun"][i] <= single.fun + 3e-5 * max(1.0, abs(single.fun)), (region, year, r["fun"][i], single.fun)

                def one(th, own):
                    vals = [gp.nlml(t, grad="exact") for t in th]
                    return np.array([v for v, _ in vals]), np.array([gg for _, gg in vals])

                alone = newton_lockstep(one, x0[None], maxiter=40, ftol=1e-12)
            assert abs(alone["fun"][0] - r["fun"][i]) <= 3e-5 * max(1.0, abs(r["fun"][i])), (region, year, alone["fun"], r["fun"][i])
            # (x itself is not pinned along the flat l -> 0 / l -> inf directions of these likelihoods: same value, both stationary)
            assert np.max(np.abs(O.mlii(alone["x"][0], X, y, M=M, grad="exact")[1])) <= 1e-3 * max(1.0, abs(fo))
            checked += 1
    assert checked == 3 * (fmax - fmin + 1)


# ---- what the driver times: the bench's own step shapes ----------------------------------------------------------------------------
TOL_PRED = 1e-8


def _bench_years(n, d, years):
    import bench
    Xb = np.zeros((years, n, d)); yb = np.zeros((years, n)); Xsb = np.zeros((years, 1, d))
    for b in range(years):
        Xb[b], yb[b], Xsb[b] = bench.synthetic_problem(n, d, 20240002 + b, m=1)
    return Xb, yb, Xsb


@pytest.mark.timeout(1200)
def test_bench_configuration_lockstep_g160_against_oracle(S):
    """configs[2] exactly as the default ``python bench.py`` times it: ONE lockstep step of G = 160 members -- the 40 years at 4 consecutive
    grid points of ``bench.grid_point`` (the first timed step: points 4 .. 7 after one warm-up step), n = 8192, d = 8, W = 8.  One member per
    grid point against the oracle; ALL 160 members against the same fits factorised in the lockstep groups of 40 that
    test_bench_configuration_lockstep_g40_against_oracle pins (the group size is a schedule: same tile kernels, same k order)."""
    import bench
    n, d, years, G = 8192, 8, 40, 160
    Xb, yb, Xsb = _bench_years(n, d, years)
    fits = np.arange(G, 2 * G)                                   # the first timed step of the default run (warmup = 1)
    pts = [bench.grid_point(int(i // years), d, "smoke") for i in fits]
    ell = np.array([p[0] for p in pts]); sn = np.array([p[1] for p in pts])
    assert len({p for p in pts}) == 4
    with S.GPR(kernel="rbf", outer_blocks=8) as gp:
        gp.upload_batch(Xb, yb, Xsb, group=G, concurrency=1)
        r = gp.run_batch(G, G, ell, sn, concurrency=1, group=G)            # fit i uses data set i % 40, as in the bench
        r40 = [gp.run_batch(G + 40 * q, 40, ell[40 * q:40 * q + 40], sn[40 * q:40 * q + 40], concurrency=1, group=40) for q in range(4)]
    assert np.all(r["info"] == 0) and np.all(r["var"] > 0) and np.all(np.isfinite(r["nlml"])) and np.all(np.isfinite(r["mean"]))
    for q in range(4):
        for key in ("mean", "var", "nlml", "sigma_f"):
            a, b = np.asarray(r[key][40 * q:40 * q + 40]), np.asarray(r40[q][key])
            # (last bits only: which tile kernel an in-panel update gets depends on members x tiles per launch)
            assert np.max(np.abs(a - b)) <= 1e-11 * max(1.0, np.max(np.abs(b))), (q, key, np.max(np.abs(a - b)))
    for i in (0, 53, 106, 159):                                  # one member per grid point
        b = int(fits[i] % years)
        ref = O.fit_predict(Xb[b], yb[b], Xsb[b], ell[i], sn[i], kind="rbf", ref_idiom=False)
        assert rel(r["mean"][i], ref["fmean"]) <= TOL_PRED and rel(r["var"][i], ref["fvar"]) <= TOL_PRED, i
        assert rel(r["nlml"][i], ref["nlml"]) <= 1e-9 and rel(r["sigma_f"][i], ref["sigma_f"]) <= 1e-9, i


@pytest.mark.timeout(1200)
def test_bench_full_grid_every_point_at_n8192(S):
    """``--grid full``: all 400 points of l = sqrt(d) logspace(-1, 1, 20) x sn~ = logspace(-3, 1, 20) at n = 8192 (4 years per point, 40 points
    per lockstep launch of 160): every fit succeeds with finite results, and the four corners of the grid (the extremes of both ranges) agree
    with the oracle.  cond(K~) reaches ~ n / sn~ = 8e6 at the small-noise corners: predictions stay within 1e-8 of the oracle's."""
    import bench
    n, d, years, G = 8192, 8, 4, 160
    Xb, yb, Xsb = _bench_years(n, d, years)
    F = 400 * years
    pts = [bench.grid_point(int(i // years), d, "full") for i in range(F)]
    ell = np.array([p[0] for p in pts]); sn = np.array([p[1] for p in pts])
    assert len(set(pts)) == 400
    with S.GPR(kernel="rbf", outer_blocks=8) as gp:
        gp.upload_batch(Xb, yb, Xsb, group=G, concurrency=1)
        r = gp.run_batch(0, F, ell, sn, concurrency=1, group=G)
    assert np.all(r["info"] == 0), np.flatnonzero(r["info"])[:10]
    assert np.all(np.isfinite(r["mean"])) and np.all(np.isfinite(r["nlml"])) and np.all(np.isfinite(r["sigma_f"])) and np.all(r["var"] > 0)
    ells, sns = bench.grid_axes(d, "full")
    for e, s_ in ((ells[0], sns[0]), (ells[-1], sns[0]), (ells[0], sns[-1]), (ells[-1], sns[-1])):
        i = int(np.flatnonzero((ell == e) & (sn == s_))[0])
        b = i % years
        ref = O.fit_predict(Xb[b], yb[b], Xsb[b], e, s_, kind="rbf", ref_idiom=False)
        scale = max(np.max(np.abs(ref["fmean"])), np.max(np.abs(ref["KXXs"])) * np.max(np.abs(ref["alpha"])))
        assert np.max(np.abs(r["mean"][i] - ref["fmean"])) <= TOL_PRED * scale, (e, s_, r["mean"][i], ref["fmean"])
        assert np.max(np.abs(r["var"][i] - ref["fvar"])) <= TOL_PRED * np.max(np.abs(ref["kss"])), (e, s_, r["var"][i], ref["fvar"])
        assert rel(r["nlml"][i], ref["nlml"]) <= 1e-9 and rel(r["sigma_f"][i], ref["sigma_f"]) <= 1e-8, (e, s_)


def test_bench_falls_back_to_fewer_grid_points_when_hbm_is_short():
    """bench.py's memory fallback: with less free HBM than the group of 160 needs (SIGP_BENCH_FREE_BYTES overrides hipMemGetInfo) the step
    shrinks by WHOLE years-worth of fits (160 -> 80 at 60 GB with no extras) and the compact line the driver parses shows both numbers."""
    env = dict(os.environ, SIGP_BENCH_FREE_BYTES="60e9")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--no-profile"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["config"]["fits_per_step"] == 80 and line["config"]["lockstep_group_asked"] == 160, line["config"]
    assert line["value"] > 0 and line["steps"] == 1 and abs(line["value"] - 80 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert len(p.stdout.strip().splitlines()[-1]) <= 1500


def test_strip_solve_skipping_the_zero_tile_slices_is_bit_identical(S):
    """panel_strip_kernel multiplies a strip by [Mt_j0 .. Mt_jj]^T; Mt_jj = inv(L_jj) is lower triangular, and the `strip_tri` form
    (default) issues only the 16-column x 16-k tile-slices of that block that are not identically zero (its column tiles interleaved
    over the two column waves).  The skipped products are exact zeros: the factor L~ itself, nlML, sigma_f and the predictions are
    bit-identical to the full products (`strip_tri` = 0) -- a single fit at three panel widths incl. a ragged last panel, and a
    lockstep batch; the fit is also checked against the oracle."""
    for n, W in ((2300, 8), (1300, 4), (1100, 2)):
        X, y, Xs = O.synthetic_problem(n, 8, 7700 + n, m=3)
        out = []
        for tri in (1, 0):
            with S.GPR(kernel="rbf", outer_blocks=W, panel_mode="strips") as gp:
                gp.set_option("strip_tri", tri)
                gp.fit(X, y, np.sqrt(8.0), 1e-1, Xs=Xs)
                mu, var = gp.predict(Xs)
                out.append((gp.nlml_, gp.sigma_f_, mu, var, gp.L_tilde_))
        assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1])), (n, W)
        if n == 2300:
            ref = O.fit_predict(X, y, Xs, np.sqrt(8.0), 1e-1, kind="rbf", ref_idiom=False)
            assert rel(out[0][2], ref["fmean"]) <= 1e-8 and rel(out[0][0], ref["nlml"]) <= 1e-10
            assert np.max(np.abs(np.tril(out[0][4]) - ref["L_tilde"])) <= 1e-11
    n, d, B = 1500, 8, 6
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 1, d))
    for b in range(B):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 270 + b, m=1)
    res = []
    for tri in (1, 0):
        with S.GPR(kernel="rbf", outer_blocks=8, panel_mode="strips") as gp:
            gp.set_option("strip_tri", tri)
            res.append(gp.fit_batch(Xb, yb, Xsb, np.full(B, 2.5), np.logspace(-2, -1, B), concurrency=1, group=B))
    for k in ("nlml", "mean", "var", "sigma_f"):
        assert np.array_equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_diagonal_tiles_of_the_trailing_update_lower_half_only(S, dtype):
    """A diagonal 128 x 128 tile of the symmetric trailing update is only ever read below its diagonal: `diag_tiles` (default on) lets
    `syrk128_kernel` load, multiply and store the 36 of its 64 16 x 16 sub-tile pairs on or below the diagonal (row AND column tiles of a
    wave interleaved so that the four waves keep 10 / 6 / 10 / 10 pairs).  Kept pairs see the same products in the same order: L~, nlML,
    sigma_f and the predictions are bit-identical to the full tiles -- single fits at sizes that reach the 128-tile update kernel
    (more than small_tile_threshold tiles), a ragged one, both schedules, and a lockstep batch."""
    kern = "rbf" if dtype == "f64" else "matern52"
    for n, W, sched in ((4100, 8, 0), (3500, 4, 1), (2300, 8, 0)):
        X, y, Xs = O.synthetic_problem(n, 8, 8800 + n, m=2)
        out = []
        for on in (1, 0):
            with S.GPR(kernel=kern, outer_blocks=W, dtype=dtype) as gp:
                gp.set_option("diag_tiles", on)
                gp.set_option("schedule", sched)
                gp.set_option("small_tile_threshold", 1)          # every trailing update on the 128-tile kernel
                gp.set_option("tiny_tile_threshold", 1)
                gp.fit(X, y, np.sqrt(8.0), 1e-1, Xs=Xs)
                mu, var = gp.predict(Xs)
                out.append((gp.nlml_, gp.sigma_f_, mu, var) + ((gp.L_tilde_,) if dtype == "f64" else ()))
        assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1])), (n, W)
        if n == 4100 and dtype == "f64":
            ref = O.fit_predict(X, y, Xs, np.sqrt(8.0), 1e-1, kind="rbf", ref_idiom=False)
            assert rel(out[0][2], ref["fmean"]) <= 1e-8 and rel(out[0][0], ref["nlml"]) <= 1e-10
            assert np.max(np.abs(out[0][4] - ref["L_tilde"])) <= 1e-11
    if dtype == "f32":
        return
    n, d, B = 2100, 8, 6
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 1, d))
    for b in range(B):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 370 + b, m=1)
    res = []
    for on in (1, 0):
        with S.GPR(kernel="rbf", outer_blocks=4) as gp:
            gp.set_option("diag_tiles", on)
            res.append(gp.fit_batch(Xb, yb, Xsb, np.full(B, 2.5), np.logspace(-2, -1, B), concurrency=1, group=B))
    for k in ("nlml", "mean", "var", "sigma_f"):
        assert np.array_equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_ride_along_tiles_multiply_only_the_rows_in_use(S, dtype):
    """The ride-along block is 128 rows of which 1 + m are in use (y and the test points of north/June1st.py:272; the rest are zero rows).
    With at most 16 in use (`ride_tiles`, default on) its tiles in the trailing updates stage and multiply the first 16-row sub-tile only.
    Same results bit for bit as whole tiles (`ride_tiles` = 0) for m = 1, 3 and 15; m = 16 (17 rows) takes whole tiles either way; all
    against the oracle.  Lockstep batch included."""
    kern = "rbf" if dtype == "f64" else "matern52"
    n, W = 3300, 8
    for m in (1, 3, 15, 16):
        X, y, Xs = O.synthetic_problem(n, 8, 9900 + m, m=m)
        out = []
        for on in (1, 0):
            with S.GPR(kernel=kern, outer_blocks=W, dtype=dtype) as gp:
                gp.set_option("ride_tiles", on)
                gp.set_option("small_tile_threshold", 1)
                gp.set_option("tiny_tile_threshold", 1)
                if dtype == "f32" and m > 3:
                    break                                   # (the fp32 engine refines at most 3 ride-along points)
                gp.fit(X, y, np.sqrt(8.0), 1e-1, Xs=Xs)
                mu, var = gp.predict(Xs)
                out.append((gp.nlml_, gp.sigma_f_, mu, var))
        if not out:
            continue
        assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1])), m
        ref = O.fit_predict(X, y, Xs, np.sqrt(8.0), 1e-1, kind=kern, ref_idiom=False)
        tol = 1e-8 if dtype == "f64" else 1e-6
        assert rel(out[0][2], ref["fmean"]) <= tol and rel(out[0][3], ref["fvar"]) <= (tol if dtype == "f64" else 1e-5), m
    if dtype == "f32":
        return
    n, d, B = 2100, 8, 6
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 2, d))
    for b in range(B):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 470 + b, m=2)
    res = []
    for on in (1, 0):
        with S.GPR(kernel="rbf", outer_blocks=4) as gp:
            gp.set_option("ride_tiles", on)
            gp.set_option("small_tile_threshold", 1)
            res.append(gp.fit_batch(Xb, yb, Xsb, np.full(B, 2.5), np.logspace(-2, -1, B), concurrency=1, group=B))
    for k in ("nlml", "mean", "var", "sigma_f"):
        assert np.array_equal(res[0][k], res[1][k]), k
