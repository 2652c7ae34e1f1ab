"""GPU tier, round 2: the reference's own kernel batched at the reference's own size (one workgroup per fit), the
BASELINE configurations at their STATED sizes, the bench configuration itself against the oracle, and RCCL at world = 1.

Tolerances as in test_hip_parity.py (predictions <= 1e-8 relative, nlML / sigma_f <= 1e-9) unless stated.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_PRED = 1e-8


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def S():
    import seaiceextentforecasting_amd as pkg
    return pkg


# ---- a10 / a11: the reference kernel, one workgroup per fit ------------------------------------------------------------
def _ragged_sets(rng, count, nmax=128):
    sets = []
    for _ in range(count):
        n = int(rng.integers(2, nmax + 1)); N = int(rng.integers(1, 90)); m = int(rng.integers(0, 9))
        X = rng.standard_normal((n, N))
        y = X @ rng.standard_normal(N) / np.sqrt(N) + 0.3 * rng.standard_normal(n)
        Xs = rng.standard_normal((m, N)) if m else None
        sets.append((X, y, Xs))
    return sets


@pytest.mark.parametrize("expm", ["eigh", "pade"])
def test_small_batch_matches_oracle_on_ragged_sets(S, expm):
    """Random (n, N, m) per data set incl. the edges n = 2, n = 128, N = 1, m = 0 and m = 8; several (l, sn~) per set."""
    rng = np.random.default_rng(11)
    sets = _ragged_sets(rng, 14)
    sets.append((rng.standard_normal((128, 60)), rng.standard_normal(128), rng.standard_normal((8, 60))))
    sets.append((rng.standard_normal((2, 1)), rng.standard_normal(2), rng.standard_normal((1, 1))))
    sets.append((rng.standard_normal((1, 3)), rng.standard_normal(1), rng.standard_normal((2, 3))))           # a single training row: M = 0, K~ = x x^T + sn~
    sets.append((rng.standard_normal((45, 200)), rng.standard_normal(45), rng.standard_normal((1, 200))))     # reference size
    with S.GPR(kernel="netdiffusion") as gp:
        sb = S.SmallBatch(gp)
        want = []
        for X, y, Xs in sets:
            ds = sb.add_dataset(X, y, Xs)
            for ell, sn in ((1e-3, 1e-1), (0.05, 1.0), (0.7, 10.0)):
                sb.add_fit(ds, ell, sn, expm=expm)
                want.append(O.fit_predict(X, y, Xs if Xs is not None else np.zeros((1, X.shape[1])), ell, sn, kind="netdiffusion", ref_idiom=False))
        r = sb.run()
    assert np.all(r["info"] == 0)
    i = 0
    for X, y, Xs in sets:
        for _ in range(3):
            ref = want[i]
            cond = np.linalg.cond(ref["K_tilde"])
            tol = max(1e-10, 50 * cond * 2.3e-16)
            assert rel(r["sigma_f"][i], ref["sigma_f"]) <= tol and rel(r["nlml"][i], ref["nlml"]) <= max(1e-9, tol), (i, X.shape)
            assert rel(r["sigma_n"][i], ref["sigma_n"]) <= tol
            if Xs is not None:
                m = Xs.shape[0]
                scale = max(np.max(np.abs(ref["fmean"])), np.max(np.abs(ref["KXXs"])) * np.max(np.abs(ref["alpha"])))
                assert np.max(np.abs(r["mean"][i, :m] - ref["fmean"])) <= max(TOL_PRED, tol) * scale, (i, X.shape)
                assert np.max(np.abs(r["var"][i, :m] - ref["fvar"])) <= max(TOL_PRED, tol) * np.max(np.abs(ref["kss"])), (i, X.shape)
                assert np.all(np.isnan(r["mean"][i, m:]))
            i += 1


def test_small_batch_isolates_a_non_spd_fit(S):
    """A singular K~ (duplicate rows, sn~ = 0) reports its LAPACK pivot and +inf / NaN (north/June1st.py:254-256);
    its neighbours in the launch are unaffected."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((40, 6)); y = rng.standard_normal(40); Xs = rng.standard_normal((1, 6))
    Xbad = X.copy(); Xbad[30:38] = Xbad[2:10]
    with S.GPR(kernel="netdiffusion") as gp:
        r = gp.fit_batch([X, Xbad, X], [y, y, y], [Xs, Xs, Xs], [0.05, 0.05, 0.05], [1e-2, 0.0, 1e-2])
    assert r["info"][0] == 0 and r["info"][2] == 0 and 1 <= r["info"][1] <= 40
    assert np.isinf(r["nlml"][1]) and np.isnan(r["mean"][1]).all() and np.isnan(r["var"][1]).all()
    ref = O.fit_predict(X, y, Xs, 0.05, 1e-2, kind="netdiffusion", ref_idiom=False)
    assert rel(r["mean"][0], ref["fmean"]) <= TOL_PRED and rel(r["mean"][2], ref["fmean"]) <= TOL_PRED
    with pytest.raises(ValueError):
        with S.GPR(kernel="netdiffusion") as gp:
            S.SmallBatch(gp).add_dataset(np.zeros((129, 3)), np.zeros(129))
    with S.GPR(kernel="netdiffusion") as gp:           # predict_batch: predictions only, LinAlgError where the reference would raise
        mu, var = gp.predict_batch([X, X], [y, 2 * y], [Xs, Xs], [0.05, 0.05], [1e-2, 1e-2])
        assert rel(mu[0], ref["fmean"]) <= TOL_PRED and rel(mu[1], 2 * ref["fmean"]) <= TOL_PRED and rel(var[0], ref["fvar"]) <= TOL_PRED
        with pytest.raises(np.linalg.LinAlgError):
            gp.predict_batch([X, Xbad], [y, y], [Xs, Xs], [0.05, 0.05], [1e-2, 0.0])


def test_reference_kernel_batch_mixes_small_and_large_orders(S):
    """fit_batch(kernel='netdiffusion') with data sets either side of the one-workgroup limit (n <= 128): the small ones go through
    the batched kernel, the large ones one at a time through the blocked engine, results in the caller's order; a non-SPD large
    member reports its pivot like the small ones do."""
    rng = np.random.default_rng(8)
    sets = []
    for n, N in ((40, 10), (300, 12), (128, 20), (129, 6)):
        X = rng.standard_normal((n, N)); y = X @ rng.standard_normal(N) / np.sqrt(N) + 0.3 * rng.standard_normal(n)
        sets.append((X, y, rng.standard_normal((2, N))))
    Xbad = sets[1][0].copy(); Xbad[200:260] = Xbad[10:70]
    sets.append((Xbad, sets[1][1], sets[1][2]))
    ell = [0.05, 0.05, 0.2, 0.05, 0.05]; sn = [1e-1, 1e-1, 1.0, 1e-2, 0.0]
    with S.GPR(kernel="netdiffusion") as gp:
        r = gp.fit_batch([q[0] for q in sets], [q[1] for q in sets], [q[2] for q in sets], ell, sn)
    assert list(r["info"][:4]) == [0, 0, 0, 0] and r["info"][4] > 0 and np.isinf(r["nlml"][4])
    for i in range(4):
        ref = O.fit_predict(sets[i][0], sets[i][1], sets[i][2], ell[i], sn[i], kind="netdiffusion", ref_idiom=False)
        assert rel(r["mean"][i], ref["fmean"]) <= TOL_PRED and rel(r["var"][i], ref["fvar"]) <= TOL_PRED, i
        assert rel(r["nlml"][i], ref["nlml"]) <= 1e-9 and rel(r["sigma_f"][i], ref["sigma_f"]) <= 1e-9, i


def test_golden_scripts_through_the_batched_engine(S, golden):
    """All 14 reference scripts with every (region, year) fit in ONE launch (retro) / every region in one launch
    (operational): same acceptance as the fit-at-a-time test -- <= 1e-8, rounded outputs equal."""
    g = golden
    script = g["script"].replace("_retro", "")
    with S.GPR(kernel="netdiffusion") as gp:
        if g["kind"] == "retro":
            out = S.retro_forecast(script, g["SIC"], g["SIEs_dt"], g["SIEs_trend"], g["args"][0], g["args"][1], SST=g["SST"], gp=gp, batched=True)
            for key, val in g["GPR"].items():
                assert np.max(np.abs(out[key] - val)) <= 1.0000001e-3, key
                assert np.mean(out[key] == val) >= 0.9, key
        recs = g["records"]
        r = gp.fit_batch([q["X"] for q in recs], [q["y"] for q in recs], [q["Xs"] for q in recs], [float(q["ell"]) for q in recs],
                         [float(q["sn_tilde"]) for q in recs], M=[q["M"] for q in recs])
        assert np.all(r["info"] == 0)
        for i, q in enumerate(recs):
            cond = np.linalg.cond(q["L_tilde"]) ** 2
            tol = max(TOL_PRED, 100 * cond * 2.3e-16)
            kss = float(q["KXsXs"][0][0])
            # the retro captures hold the rounded outputs only: form fmean / fvar from the captured locals (north/June1st.py:276-277)
            fmean = float(q["fmean"]) if "fmean" in q else float((q["KXXs"].T @ q["alpha"])[0, 0])
            fvar = float(q["fvar"]) if "fvar" in q else float((q["KXsXs"] - q["v"].T @ q["v"])[0, 0])
            assert abs(r["mean"][i, 0] - fmean) <= tol * max(abs(fmean), np.abs(q["KXXs"]).max() * np.abs(q["alpha"]).max())
            assert abs(r["var"][i, 0] - fvar) <= tol * kss
            assert rel(r["sigma_f"][i], q["sigma_f"]) <= max(1e-10, tol)
            n = q["y"].shape[0]
            nl_ref = float((q["y"].T @ q["alpha"])[0, 0]) / 2 + np.log(np.diag(q["L"])).sum() + n * np.log(2 * np.pi) / 2
            assert abs(r["nlml"][i] - nl_ref) <= 1e-9 * abs(nl_ref)


def test_reference_grid_search_in_one_launch(S, golden):
    """a11: the reference's own 20 x 20 grid (logspace(-7,2,20) x logspace(-3,9,20), north/June1st.py:210-211) for a golden
    (region, year) in one launch, against the reference's live MLII closure values where captured and the oracle's nlML on
    a sub-grid."""
    q = golden["records"][0]
    X, y, M = q["X"], q["y"], q["M"]
    with S.GPR(kernel="netdiffusion", expm="eigh") as gp:
        G = gp.nlml_grid(X, y, S.LGRID, S.SGRID, M=M)
    assert G.shape == (20, 20)
    for i in range(0, 20, 3):
        for j in range(0, 20, 3):
            if S.LGRID[i] * np.abs(M).max() > 50:
                continue                                    # expm loses digits there; eigh does not
            nl, _ = O.mlii(np.log([S.LGRID[i], S.SGRID[j]]), X, y, kind="netdiffusion", M=M, grad="ref")
            if np.isinf(nl):
                assert np.isinf(G[i, j])
            else:
                assert abs(G[i, j] - float(nl)) <= 1e-8 * max(1.0, abs(float(nl))), (i, j, G[i, j], nl)


def test_retro_grid_search_shapes_and_table_entry(S, golden):
    """retro_grid_search: 3 regions x years x 400 fits in one call; the script's own table entry (when it sits on the grid)
    gives the nlML of the fit the forecast used."""
    g = golden
    if g["kind"] != "retro":
        pytest.skip("retro scripts only")
    script = g["script"].replace("_retro", "")
    fmin, fmax = g["args"][0], g["args"][1]
    res = S.retro_grid_search(script, g["SIC"], g["SIEs_dt"], fmin, fmax, SST=g["SST"])
    tab = S.SCRIPT_TABLE[script]
    assert set(res) == set(tab["regions"])
    for k, region in enumerate(tab["regions"]):
        assert res[region].shape == (fmax - fmin + 1, 20, 20)
        li = np.flatnonzero(np.isclose(S.LGRID, tab["ell"][k], rtol=1e-12)); si = np.flatnonzero(np.isclose(S.SGRID, tab["sn"][k], rtol=1e-12))
        if len(li) == 0 or len(si) == 0 or tab["ell"][k] > 1:
            continue
        for q in g["records"]:
            if int(q["k"]) != k:
                continue
            n = q["y"].shape[0]
            nl_ref = float((q["y"].T @ q["alpha"])[0, 0]) / 2 + np.log(np.diag(q["L"])).sum() + n * np.log(2 * np.pi) / 2
            assert abs(res[region][int(q["year"]) - fmin, li[0], si[0]] - nl_ref) <= 1e-8 * max(1.0, abs(nl_ref))


# ---- the BASELINE configurations at their stated sizes ------------------------------------------------------------------
@pytest.mark.timeout(1200)
def test_bench_configuration_lockstep_g40_against_oracle(S):
    """configs[2] exactly as bench.py runs it: one lockstep step of G = 40 members, W = 8, strip panels, n = 8192, d = 8.
    Four members (first, two in the middle, last) against the oracle; every member through identities."""
    n, d, G = 8192, 8, 40
    Xb = np.zeros((G, n, d)); yb = np.zeros((G, n)); Xsb = np.zeros((G, 1, d))
    for b in range(G):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 20240002 + b, m=1)
    ell = np.full(G, np.sqrt(d) * 10 ** (-1 / 3)); sn = np.full(G, 10 ** (-3 + 4 / 3))
    with S.GPR(kernel="rbf", outer_blocks=8) as gp:
        gp.upload_batch(Xb, yb, Xsb, group=G, concurrency=1)
        r = gp.run_batch(0, G, ell, sn, concurrency=1, group=G)
    assert np.all(r["info"] == 0) and np.all(r["var"] > 0) and np.all(np.isfinite(r["nlml"]))
    for b in (0, 13, 27, 39):
        ref = O.fit_predict(Xb[b], yb[b], Xsb[b], ell[b], sn[b], kind="rbf", ref_idiom=False)
        assert rel(r["mean"][b], ref["fmean"]) <= TOL_PRED and rel(r["var"][b], ref["fvar"]) <= TOL_PRED, b
        assert rel(r["nlml"][b], ref["nlml"]) <= 1e-9 and rel(r["sigma_f"][b], ref["sigma_f"]) <= 1e-9, b


@pytest.mark.timeout(1200)
def test_config3_n16384_d16_single_gpu_and_sharded_protocol(S):
    """configs[3] at its stated size (n = 16384, d = 16 fp64 RBF): GPR.fit and the sharded panel protocol at world = 1,
    both against the oracle (m = 4 test points, nlML, sigma_f) and through size-independent identities."""
    n, d = 16384, 16
    X, y, Xs = O.synthetic_problem(n, d, 20240003, m=4)
    ell, sn = np.sqrt(d), 1e-2
    ref = O.fit_predict(X, y, Xs, ell, sn, kind="rbf", ref_idiom=False)
    with S.GPR(kernel="rbf") as gp:
        gp.fit(X, y, ell, sn, Xs=Xs)
        mu, var = gp.predict(Xs)
        assert rel(mu, ref["fmean"]) <= TOL_PRED and rel(var, ref["fvar"]) <= TOL_PRED
        assert rel(gp.nlml_, ref["nlml"]) <= 1e-9 and rel(gp.sigma_f_, ref["sigma_f"]) <= 1e-9
        at = gp.alpha_[:, 0] * gp.sigma_f_
        assert abs(float(y @ at) - n * gp.sigma_f_) <= 1e-9 * n * gp.sigma_f_            # y^T alpha~ = n sigma_f
        rows = np.random.default_rng(0).choice(n, 64, replace=False)
        Kr = O.cov_unit("rbf", X[rows], X, ell); Kr[np.arange(64), rows] += sn
        assert np.max(np.abs(Kr @ at - y[rows])) <= 1e-8 * np.max(np.abs(y))            # K~ alpha~ = y on sampled rows
        assert rel(at, ref["A_tilde"][:, 0]) <= 1e-8
        nl1 = gp.nlml_
    with S.DistributedGPR("rbf", 0, 1, None, device=0, outer_blocks=8) as dg:      # sigp_dist_fit, one rank (world 2 at this size: tests/test_sharded.py)
        dg.fit(X, y, ell, sn, Xs=Xs)
        mu2, var2 = dg.predict(Xs)
        assert rel(mu2, ref["fmean"]) <= TOL_PRED and rel(var2, ref["fvar"]) <= TOL_PRED
        assert rel(dg.nlml_, ref["nlml"]) <= 1e-9 and abs(dg.nlml_ - nl1) <= 1e-12 * abs(nl1)


@pytest.mark.timeout(1200)
def test_config4_n32768_d32_fp32_matern_with_refinement(S):
    """configs[4] at its stated size on one GPU: fp32 Matern-5/2 factor + fp64 iterative refinement.  Stated tolerances:
    refinement residual max|y - K~ alpha~| / max|y| <= 1e-10, mean vs an fp64 fit of the same problem on the HIP engine
    <= 1e-6 relative, sigma_f <= 1e-6, ride-along variance <= 1e-5."""
    n, d = 32768, 32
    X, y, Xs = O.synthetic_problem(n, d, 20240004, m=2)
    ell, sn = np.sqrt(d), 1e-1
    with S.GPR(kernel="matern52") as g64:
        g64.fit(X, y, ell, sn, Xs=Xs)
        mu64, var64 = g64.predict(Xs)
        sf64, nl64 = g64.sigma_f_, g64.nlml_
        a64 = g64.alpha_[:, 0]
    rows = np.random.default_rng(0).choice(n, 32, replace=False)
    Kr = O.cov_unit("matern52", X[rows], X, ell); Kr[np.arange(32), rows] += sn
    assert np.max(np.abs(Kr @ (a64 * sf64) - y[rows])) <= 1e-9 * np.max(np.abs(y))     # the fp64 fit is itself consistent with the oracle's covariance
    with S.GPR(kernel="matern52", dtype="f32") as g32:
        g32.fit(X, y, ell, sn, Xs=Xs)
        mu, var = g32.predict(Xs)
        assert 0.0 < g32.refine_residual_ <= 1e-10, g32.refine_residual_      # an exactly-zero fp64 residual on 32768 rows would mean "not measured"
        assert rel(mu, mu64) <= 1e-6 and rel(g32.sigma_f_, sf64) <= 1e-6, (rel(mu, mu64), rel(g32.sigma_f_, sf64))
        assert rel(var, var64) <= 1e-5 and rel(g32.nlml_, nl64) <= 1e-5
        a32 = g32.alpha_[:, 0]
        assert rel(a32, a64) <= 1e-6
        assert np.max(np.abs(Kr @ (a32 * g32.sigma_f_) - y[rows])) <= 1e-9 * np.max(np.abs(y))
        sf32, nl32 = g32.sigma_f_, g32.nlml_
    # ... and against the ORACLE itself at the stated size, not only against the HIP engine's own fp64 fit: its numbers for exactly
    # these inputs are the fixture tests/golden/config4_oracle.npz (tests/golden/make_golden_config4.py: fit_predict_lean, ~3 min
    # of host time); SIGP_LIVE_ORACLE=1 recomputes them here instead
    if os.environ.get("SIGP_LIVE_ORACLE") == "1":
        ref = O.fit_predict_lean(X, y, Xs, ell, sn, kind="matern52", threads=8)
        ref["A_tilde"] = ref["A_tilde"][:, 0]
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "config4_oracle.npz"))
        assert (int(z["n"]), int(z["d"]), int(z["seed"]), int(z["m"])) == (n, d, 20240004, 2) and float(z["ell"]) == ell and float(z["sn"]) == sn
        ref = {k: z[k] for k in ("fmean", "fvar", "sigma_f", "nlml", "A_tilde")}
    assert rel(mu, ref["fmean"]) <= 1e-6 and rel(var, ref["fvar"]) <= 1e-5, (rel(mu, ref["fmean"]), rel(var, ref["fvar"]))
    assert rel(sf32, ref["sigma_f"]) <= 1e-6 and rel(nl32, ref["nlml"]) <= 1e-5
    assert rel(a32 * sf32, ref["A_tilde"]) <= 1e-6
    assert rel(mu64, ref["fmean"]) <= TOL_PRED and rel(var64, ref["fvar"]) <= TOL_PRED and rel(nl64, ref["nlml"]) <= 1e-9    # the fp64 engine at n = 32768, d = 32


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,n,d", [("rbf", 4096, 8), ("matern52", 1500, 5), ("netdiffusion", 2100, 12)])
def test_mlii_gradient_at_multi_block_sizes(S, kind, n, d):
    """MLII (north/June1st.py:235-257) at sizes where the recursive triangular inversion has several levels and a ragged
    tail (n_pad/128 = 32, 12, 17): value and exact gradient against the oracle; for the reference kernel also the
    reference's own formulae (grad='ref')."""
    X, y, _ = O.synthetic_problem(n, d, 4096 + n, m=1)
    th = np.log([np.sqrt(d), 1e-2]) if kind != "netdiffusion" else np.log([0.05, 0.3])
    with S.GPR(kernel=kind) as gp:
        gp.set_data(X, y)
        modes = ("exact", "ref") if kind == "netdiffusion" else ("exact",)
        for mode in modes:
            f0, g0 = gp.nlml(th, grad=mode)
            fo, go = O.mlii(th, X, y, kind=kind, grad=mode)
            assert abs(f0 - fo) <= 1e-9 * abs(fo), (mode, f0, fo)
            assert np.allclose(g0, go, rtol=1e-6, atol=1e-7 * max(1.0, np.max(np.abs(go)))), (mode, g0, go)
        # a fit after a gradient call still predicts correctly (the gradient has its own workspaces)
        gp.refit(float(np.exp(th[0])), float(np.exp(th[1])))
        Xs = X[:3] + 0.01
        gp.nlml(th, grad="exact")
        gp.refit(float(np.exp(th[0])), float(np.exp(th[1])))
        mu, var = gp.predict(Xs)
    ref = O.fit_predict(X, y, Xs, float(np.exp(th[0])), float(np.exp(th[1])), kind=kind, ref_idiom=False)
    assert rel(mu, ref["fmean"]) <= TOL_PRED and rel(var, ref["fvar"]) <= max(TOL_PRED, 1e-6 * float(np.exp(th[1])))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,n,d,B", [("rbf", 1500, 8, 5), ("matern52", 1100, 5, 3), ("rbf", 4096, 8, 4), ("rbf", 300, 3, 7)])
def test_lockstep_mlii_gradients_match_the_oracle_for_every_member(S, kind, n, d, B):
    """sigp_nlml_grad_batch: value and exact gradient of the profiled nlML (north/June1st.py:235-257 with the true derivative) for a
    lockstep group -- every member against oracle.mlii(grad='exact'), against the single-fit entry point, with ragged group tails
    and a non-SPD member isolated."""
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n))
    for b in range(B):
        Xb[b], yb[b], _ = O.synthetic_problem(n, d, 900 + 17 * b + n, m=1)
    rng = np.random.default_rng(n)
    theta = np.column_stack([np.log(np.sqrt(d)) + 0.3 * rng.standard_normal(B), np.log(1e-2) + 0.5 * rng.standard_normal(B)])
    with S.GPR(kernel=kind) as gp:
        gp.upload_batch(Xb, yb, None, group=3, concurrency=1)
        for group in (3, B):                               # a ragged last group, and all members in one
            val, g = gp.nlml_batch(theta, grad="exact", group=group)
            for b in range(B if n <= 1500 else 2):
                rv, rg = O.mlii(theta[b], Xb[b], yb[b], kind=kind, grad="exact")
                assert abs(val[b] - rv) <= 1e-9 * abs(rv), (b, val[b], rv)
                assert np.max(np.abs(g[b] - rg)) <= 1e-6 * max(1.0, np.max(np.abs(rg))), (b, g[b], rg)
        v0, _ = gp.nlml_batch(theta, grad=None, group=B)
        assert np.array_equal(v0, val)                     # value-only mode: the same factorisation
        gp.set_data(Xb[1], yb[1])
        v1, g1 = gp.nlml(theta[1], grad="exact")           # the single-fit entry point on member 1
        assert abs(v1 - val[1]) <= 1e-12 * abs(v1) and np.max(np.abs(g1 - g[1])) <= 1e-8 * max(1.0, np.max(np.abs(g1)))
        # a member whose K~ is not SPD (duplicated rows, sn~ -> 0) is isolated: +inf for it, finite for the others
        Xbad = Xb.copy(); Xbad[1, 40:60] = Xbad[1, 100:120]
        gp.upload_batch(Xbad, yb, None, group=B, concurrency=1)
        tb = theta.copy(); tb[1, 1] = -80.0
        val, g = gp.nlml_batch(tb, grad="exact", group=B)
        assert np.isinf(val[1]) and np.all(np.isinf(g[1])) and np.all(np.isfinite(np.delete(val, 1))) and np.all(np.isfinite(np.delete(g, 1, axis=0)))


@pytest.mark.timeout(900)
def test_batched_multi_start_optimiser_over_the_years(S):
    """GPR.optimize_batch: BFGS on (log l, log sn~) for every data set at once, one lockstep device call per round -- reaches the
    optimum scipy's L-BFGS-B finds for each data set through the single-fit MLII, in far fewer device calls than the sum of theirs."""
    B, n, d = 6, 700, 4
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n))
    for b in range(B):
        Xb[b], yb[b], _ = O.synthetic_problem(n, d, 3100 + b, m=1)
    th0 = np.log([np.sqrt(d), 1e-1])
    with S.GPR(kernel="rbf") as gp:
        r = gp.optimize_batch(Xb, yb, th0, group=B, maxiter=40)
        assert np.all(r["converged"]), r
        calls = 0
        for b in range(B):
            gp.set_data(Xb[b], yb[b])
            ref = gp.optimize(th0, method="L-BFGS-B")
            calls += ref.nfev
            assert r["fun"][b] <= ref.fun + 1e-6 * abs(ref.fun), (b, r["fun"][b], ref.fun)
            assert np.max(np.abs(r["x"][b] - ref.x)) <= 5e-3, (b, r["x"][b], ref.x)
            # the optimum is a stationary point of the ORACLE's profiled nlML as well
            _, og = O.mlii(r["x"][b], Xb[b], yb[b], kind="rbf", grad="exact")
            assert np.max(np.abs(og)) <= 1e-2 * max(1.0, abs(r["fun"][b]) * 1e-3), (b, og)
        assert r["nfev"] < calls


def test_rejected_experiments_are_not_in_the_product_library(S):
    """The tile-walk / tile-shape experiments of DESIGN.md section 7 (all measured slower) live in libsigp_debug.so only: the product
    library refuses their switches (tools/tile_opt_check.py checks their bit-identity against the debug library)."""
    with S.GPR(kernel="rbf") as gp:
        for name in ("xcd_chunks", "update_wgs", "update_late", "pipeline_head", "wide_tiles", "n64_tiles", "patch", "small_nt64", "reserve_cus", "panel_ll"):
            with pytest.raises(ValueError, match="libsigp_debug"):
                gp.set_option(name, 1)
        for name, v in (("outer_blocks", 4), ("lookahead", 1), ("panel_chain", 15), ("first_on_panel", 1), ("strips_after_update", 0), ("schedule", 0)):
            gp.set_option(name, v)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_panel_chain_and_first_update_placement_are_schedules_only(S, dtype):
    """The latency-chain form of a panel (right-looking column by column, the other columns' update riding in the diagonal-block
    launch: diag_update_kernel) and the placement of the next panel's first update (panel stream / update stream) change which
    launch and which stream a tile's update runs in, never its k order: factor, nlML and predictions bit-identical to the
    recursive panel -- single fits of several shapes (ragged last panel, one-panel matrix, ride rows) and a small lockstep
    group.  The single fit is also checked against the oracle."""
    kern = "rbf" if dtype == "f64" else "matern52"
    for n, W in ((2300, 8), (1100, 4), (700, 16), (4100, 8)):
        X, y, Xs = O.synthetic_problem(n, 8, 900 + n, m=2)
        out = []
        for chain, first in ((0, 0), (1, 1), (1, 2), (3, 0), (0, 2), (5, 1), (7, 2), (15, 0), (8, 1)):      # bit 2: the fused chain link (chain_link_kernel)
            with S.GPR(kernel=kern, outer_blocks=W, dtype=dtype) as gp:
                gp.set_option("panel_chain", chain)
                gp.set_option("first_on_panel", first)
                gp.fit(X, y, np.sqrt(8.0), 1e-1, Xs=Xs)
                mu, var = gp.predict(Xs)
                out.append((gp.nlml_, gp.sigma_f_, mu, var) + ((gp.L_tilde_,) if dtype == "f64" else ()))
        for o in out[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(out[0], o)), (n, W)
        if n == 2300 and dtype == "f64":
            ref = O.fit_predict(X, y, Xs, np.sqrt(8.0), 1e-1, kind="rbf", ref_idiom=False)
            assert rel(out[1][2], ref["fmean"]) <= TOL_PRED and rel(out[1][0], ref["nlml"]) <= 1e-10
    if dtype == "f32":
        return
    n, d, B = 1500, 8, 6
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 1, d))
    for b in range(B):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 170 + b, m=1)
    res = []
    for chain, mode in ((0, "recursive"), (1, "recursive"), (3, "strips"), (0, "strips"), (5, "recursive"), (7, "strips")):
        with S.GPR(kernel="rbf", outer_blocks=4, panel_mode=mode) as gp:
            gp.set_option("panel_chain", chain)
            res.append(gp.fit_batch(Xb, yb, Xsb, np.full(B, 2.5), np.logspace(-2, -1, B), concurrency=1, group=B))
    for k in ("nlml", "mean", "var", "sigma_f"):
        assert np.array_equal(res[0][k], res[1][k]) and np.array_equal(res[2][k], res[3][k]), k
        assert np.array_equal(res[0][k], res[4][k]) and np.array_equal(res[2][k], res[5][k]), k


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_predict_many_points_in_lockstep_groups(S, kind):
    """north/June1st.py:272-277 for many test points: more than 128 points go through the forward solve in groups of up to 16
    chunks advancing in lockstep (sigp_predict).  Same numbers as chunk-by-chunk prediction (bit-identical: a chunk's rows
    never mix with another's) and as the oracle; a ragged last chunk and a ragged last group included."""
    n, d, m = 1500, 6, 16 * 128 + 300
    X, y, _ = O.synthetic_problem(n, d, 4242, m=1)
    Xs = np.random.default_rng(7).standard_normal((m, d))
    with S.GPR(kernel=kind) as gp:
        gp.fit(X, y, 2.0, 5e-2)
        mu, var = gp.predict(Xs)
        mu1 = np.concatenate([gp.predict(Xs[i:i + 128])[0] for i in range(0, m, 128)])
        var1 = np.concatenate([gp.predict(Xs[i:i + 128])[1] for i in range(0, m, 128)])
    assert np.array_equal(mu, mu1) and np.array_equal(var, var1)
    ref = O.fit_predict(X, y, Xs, 2.0, 5e-2, kind=kind, ref_idiom=False)
    assert rel(mu, ref["fmean"]) <= TOL_PRED and rel(var, ref["fvar"]) <= TOL_PRED


def test_randomised_fits_against_the_oracle(S):
    """60 seeded random single fits (orders on and around the 16 / 128 / 1024 block boundaries, 1..40 features, 0..4 ride rows,
    both covariances, fp64 and fp32, outer panel widths 1..16, every panel / first-update schedule, look-ahead on and off)
    against the oracle: fp64 within 1e-8 (mean, variance, sigma_f) / 1e-9 (nlML), fp32 within its stated tolerances
    (tools/fuzz_fits.py; 500 cases of it ran clean while the round's kernels were written)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_fits", os.path.join(ROOT, "tools", "fuzz_fits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fails, worst = mod.run(60, 20241004, verbose=False)
    assert fails == 0, worst
    # lockstep batches: 1..4 data sets x 2..10 fits in groups of 1..F (ragged last group), strips forced on small shapes, a
    # singular member now and then (150 cases of it ran clean)
    fails, worst = mod.run_batches(15, 20241005, verbose=False)
    assert fails == 0, worst


def test_diagonal_block_kernel_reports_the_first_bad_pivot(S):
    """np.linalg.cholesky raises on the first non-positive pivot (north/June1st.py:265 inside MLII's try): the diagonal-block
    kernel keeps LAPACK's info = 1-based index of that pivot, whichever 16-column step, pivot wave or block it falls in."""
    rng = np.random.default_rng(5)
    n = 400
    for bad in (1, 16, 17, 100, 128, 129, 257, 400):
        B = rng.standard_normal((n, n)); K = B @ B.T + n * np.eye(n)
        # the leading minor of order `bad` gets a pivot of -1: K[bad-1, bad-1] = (its Schur-complement part) - 1
        Lk = np.linalg.cholesky(K[:bad - 1, :bad - 1]) if bad > 1 else np.zeros((0, 0))
        v = np.linalg.solve(Lk, K[:bad - 1, bad - 1]) if bad > 1 else np.zeros(0)
        K[bad - 1, bad - 1] = float(v @ v) - 1.0
        # K~ = X Sigma X^T + 0 I with X = I and "Sigma" = K: the C entry point factors exactly this matrix
        from seaiceextentforecasting_amd import _lib as L
        with S.GPR(kernel="netdiffusion") as gp:
            gp.set_data(np.eye(n), rng.standard_normal(n), M=np.zeros((n, n)))
            out = np.zeros(4); mean = np.zeros(1); var = np.zeros(1)
            rc = gp._lib.sigp_fit_predict(gp._h, gp._kid, 1.0, 0.0, L.ptr(np.ascontiguousarray(K)), n, L.ptr(out), L.ptr(mean), L.ptr(var))
        assert rc == L.NOT_SPD and int(out[2]) == bad, (bad, rc, out)


def test_fp32_refinement_from_the_stored_fp64_matrix_equals_the_recomputed_one(S):
    """fp32 engine (BASELINE configs[4]): the covariance build writes K~ in fp64 beside the fp32 matrix and the refinement's residuals
    r = b - K~ x read its lower triangle -- once (kres_sym_kernel, option refine_sym, default) or in two coalesced passes
    (kres_lower_cols / _rows) -- instead of recomputing n^2 covariances per residual (option refine_stored, default on).  Both ways refine the same systems to fp64 accuracy: solutions, nlML
    and predictions agree to 1e-11 and with the oracle at the fp32 engine's tolerances -- ragged orders (chunk and block boundaries of
    both passes), 0..3 ride points, both kernels, a lockstep group, and one handle reused for a smaller problem after a larger one."""
    cases = [(37, 3, 0, "rbf"), (255, 5, 1, "matern52"), (256, 4, 2, "rbf"), (257, 9, 3, "matern52"), (1000, 8, 2, "rbf"), (2311, 32, 1, "matern52"),
             (4100, 16, 2, "matern52"), (700, 6, 2, "rbf"), (8300, 8, 0, "rbf")]
    with S.GPR(kernel="rbf", dtype="f32") as keep_rbf, S.GPR(kernel="matern52", dtype="f32") as keep_mat:
        for n, d, m, kind in cases:
            X, y, Xs = O.synthetic_problem(n, d, 4000 + n, m=max(m, 1))
            Xs = Xs[:m] if m else None
            ell, sn = float(np.sqrt(d)), 0.1
            got = []
            for stored, sym in ((1, 1), (0, 1), (1, 0)):     # one pass over the stored triangle (kres_sym_kernel, default) / recomputed / two passes
                gp = keep_mat if kind == "matern52" else keep_rbf
                gp.set_option("refine_stored", stored)
                gp.set_option("refine_sym", sym)
                gp.fit(X, y, ell, sn, Xs=Xs)
                mu, var = gp.predict(Xs) if m else (np.zeros(0), np.zeros(0))
                got.append((gp.alpha_.copy(), gp.nlml_, gp.sigma_f_, mu, var, gp.refine_residual_))
            a1, a0, a2 = got
            for ax in (a0, a2):
                assert rel(a1[0], ax[0]) <= 1e-11 and rel(a1[1], ax[1]) <= 1e-12 and rel(a1[2], ax[2]) <= 1e-12, (n, kind)
                if m:
                    assert rel(a1[3], ax[3]) <= 1e-11 and rel(a1[4], ax[4]) <= 1e-10, (n, kind)
                assert 0 <= ax[5] <= 1e-10
            assert 0 <= a1[5] <= 1e-10
            ref = O.fit_predict(X, y, Xs if m else X[:1], ell, sn, kind=kind, ref_idiom=False)
            assert rel(a1[0], ref["alpha"]) <= 1e-6 and rel(a1[1], ref["nlml"]) <= 5e-5, (n, kind)
            if m:
                assert rel(a1[3], ref["fmean"][:m]) <= 1e-6 and rel(a1[4], ref["fvar"][:m]) <= 1e-5, (n, kind)
    n, d, B = 1300, 7, 5
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 2, d))
    for b in range(B):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 820 + b, m=2)
    ell = np.linspace(2.0, 3.0, B); sn = np.full(B, 0.1)
    res = []
    for stored in (1, 0):
        with S.GPR(kernel="matern52", dtype="f32") as gp:
            gp.set_option("refine_stored", stored)
            res.append(gp.fit_batch(Xb, yb, Xsb, ell, sn, concurrency=1, group=3))      # groups of 3 and 2 members
    for k in ("mean", "var", "nlml", "sigma_f"):
        assert rel(res[0][k], res[1][k]) <= 1e-10, k


def test_fp32_lockstep_batch_matches_oracle(S):
    """fp32 engine, batch path: the factorisations of a group run in lockstep (one build, one blocked Cholesky over all
    members), the refinement member by member.  Different data sets and hyper-parameters per member, a non-SPD member
    isolated, a ragged last group; tolerances of the fp32 engine (mean / sigma_f <= 1e-6, nlML <= 1e-5 vs the fp64 oracle)."""
    n, d, B, F = 900, 6, 3, 7
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 2, d))
    for b in range(B):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 640 + b, m=2)
    Xb[1, 500:560] = Xb[1, 100:160]                    # data set 1 is singular without noise
    ell = np.array([2.0, 2.5, 3.0, 2.2, 2.0, 2.8, 2.4]); sn = np.array([1e-1, 1e-1, 2e-1, 1e-1, 0.0, 1e-1, 3e-1])
    with S.GPR(kernel="matern52", dtype="f32") as gp:
        res = {}
        for group in (4, 1):
            res[group] = gp.fit_batch(Xb, yb, Xsb, ell, sn, concurrency=1, group=group)
        r = res[4]
        assert r["info"][4] > 0 and np.isinf(r["nlml"][4]) and np.all(np.delete(r["info"], 4) == 0)      # fit 4 = data set 1 with sn~ = 0
        for i in range(F):
            if i == 4:
                continue
            ref = O.fit_predict(Xb[i % B], yb[i % B], Xsb[i % B], ell[i], sn[i], kind="matern52", ref_idiom=False)
            assert rel(r["mean"][i], ref["fmean"]) <= 1e-6 and rel(r["sigma_f"][i], ref["sigma_f"]) <= 1e-6, i
            assert rel(r["var"][i], ref["fvar"]) <= 1e-5 and rel(r["nlml"][i], ref["nlml"]) <= 1e-5, i
        ok = np.delete(np.arange(F), 4)
        assert rel(res[4]["mean"][ok], res[1]["mean"][ok]) <= 1e-9 and rel(res[4]["nlml"][ok], res[1]["nlml"][ok]) <= 1e-9   # lockstep == one at a time
        gp.fit(Xb[0], yb[0], ell[0], sn[0], Xs=Xsb[0])       # the single-fit state still works after a batch
        mu, _ = gp.predict(Xsb[0])
        assert rel(mu, r["mean"][0]) <= 1e-9


# ---- 8f-2: the feature pipeline's tau() on the device -----------------------------------------------------------------------
@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_complex_networks_tau_on_the_device(S, case):
    """networks.Network.tau(engine=gp): correlation matrix as one fp64 MFMA product + fused threshold reduction
    (sigp_corr_tau) against the reference module's goldens: tau <= 1e-13, the areas V identical; intra_links(engine=gp)
    (sigp_area_sums): anomaly series bit-identical to the goldens (same additions in the same order)."""
    import seaiceextentforecasting_amd.networks as NW
    z = np.load(os.path.join(ROOT, "tests", "golden", "networks_%s.npz" % case), allow_pickle=False)
    data, aux, latlon = z["data"], z["aux"], bool(int(z["latlon"]))
    host = NW.Network(data=data.copy())
    NW.Network.tau(host, 0.01)
    with S.GPR(kernel="rbf") as gp:
        net = NW.Network(data=data.copy())
        NW.Network.tau(net, 0.01, engine=gp)
        assert abs(net.tau - float(z["tau"])) <= 1e-13 * abs(float(z["tau"]))
        off = ~np.eye(host._R.shape[0], dtype=bool)
        assert np.array_equal(np.isnan(net._R), np.isnan(host._R))
        assert np.nanmax(np.abs(net._R[off] - host._R[off])) <= 1e-13
        NW.Network.area_level(net, latlon_grid=latlon)
        ids = [int(i) for i in z["area_ids"]]
        assert list(net.V.keys()) == ids
        for k in ids:
            assert np.array_equal(np.array(net.V[k], dtype=np.int64), z["V/%d" % k]), k
        if latlon:
            NW.Network.intra_links(net, lat=aux, engine=gp)
        else:
            NW.Network.intra_links(net, area=aux, engine=gp)
    for k in ids:
        assert np.array_equal(net.anomaly[k], z["anomaly/%d" % k]), k
        assert np.allclose(net.links[k], z["links/%d" % k], rtol=1e-13, atol=1e-13)
    assert np.array_equal(np.isnan(net.strengthmap), np.isnan(z["strengthmap"]))


def test_tau_on_the_device_realistic_grid(S):
    """57 x 57 x 41 field (the north scripts' grid, north/June1st.py:74-75): ~2000 active cells, R = 2000 x 2000."""
    import seaiceextentforecasting_amd.networks as NW
    rng = np.random.default_rng(5)
    base = rng.standard_normal((6, 41))
    field = np.full((57, 57, 41), np.nan)
    for i in range(57):
        for j in range(57):
            if (i - 28) ** 2 + (j - 28) ** 2 < 26 ** 2:
                field[i, j] = base[(i // 20) * 2 + (j // 30)] + 0.7 * rng.standard_normal(41)
    host = NW.Network(data=field.copy()); NW.Network.tau(host, 0.01)
    with S.GPR(kernel="rbf") as gp:
        dev = NW.Network(data=field.copy()); NW.Network.tau(dev, 0.01, engine=gp)
    assert abs(dev.tau - host.tau) <= 1e-13 * abs(host.tau)
    off = ~np.eye(host._R.shape[0], dtype=bool)
    assert np.max(np.abs(dev._R[off] - host._R[off])) <= 1e-13


def test_detrend_on_the_device_matches_reference(S):
    """8f-4: callers.detrend(engine=gp) -- every cut-off year's per-pixel line removal in one launch -- against the
    reference's own detrend() captures (tests/golden/callers.npz), NaN pattern included, <= 1e-12."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "callers.npz"), allow_pickle=False)
    data = z["detrend_retro/data"]
    fmin, fmax = [int(v) for v in z["detrend_retro/args"]]
    with S.GPR(kernel="rbf") as gp:
        ds = S.detrend({"data": data.copy()}, fmin, fmax, engine=gp)
        ds2 = S.detrend({"data": data.copy()}, engine=gp)
    for year in range(fmin, fmax + 1):
        for key in ("dt_%d" % year, "trend_%d" % year):
            ref = z["detrend_retro/" + key]
            assert ds[key].shape == ref.shape and np.array_equal(np.isnan(ds[key]), np.isnan(ref)), key
            assert np.nanmax(np.abs(ds[key] - ref)) <= 1e-12 * max(1.0, np.nanmax(np.abs(ref))), key
    assert np.array_equal(np.isnan(ds2["dt"]), np.isnan(z["detrend_op/dt"]))
    assert np.nanmax(np.abs(ds2["dt"] - z["detrend_op/dt"])) <= 1e-12 and np.nanmax(np.abs(ds2["trend"] - z["detrend_op/trend"])) <= 1e-12


def test_two_handles_on_two_threads_are_independent(S):
    """sigp.h: a handle is not thread-safe, DISTINCT handles are independent.  Two threads, each with its own handle (ctypes
    releases the GIL inside the C calls), interleave fits of different kernels, sizes and batch / single paths; every result
    against the oracle.  Also exercises the once-per-device kernel-attribute setup under contention."""
    import threading
    errors = []

    def worker(seed, kind):
        try:
            rng = np.random.default_rng(seed)
            with S.GPR(kernel=kind) as gp:
                for it in range(6):
                    n = int(rng.integers(200, 1500)); d = int(rng.integers(2, 9))
                    X, y, Xs = O.synthetic_problem(n, d, 1000 * seed + it, m=2)
                    ell, sn = (float(np.sqrt(d)), 1e-2) if kind != "netdiffusion" else (0.05, 1e-1)
                    ref = O.fit_predict(X, y, Xs, ell, sn, kind=kind, ref_idiom=False)
                    if kind != "netdiffusion" and it % 2:
                        r = gp.fit_batch(X, y, Xs, [ell, ell], [sn, sn], concurrency=2, group=2)
                        mu, var, nl = r["mean"][1], r["var"][1], r["nlml"][1]
                    else:
                        gp.fit(X, y, ell, sn, Xs=Xs)
                        mu, var = gp.predict(Xs); nl = gp.nlml_
                    assert rel(mu, ref["fmean"]) <= TOL_PRED and rel(var, ref["fvar"]) <= TOL_PRED and rel(nl, ref["nlml"]) <= 1e-9, (seed, it)
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((seed, kind, repr(e)))

    ts = [threading.Thread(target=worker, args=(1, "rbf")), threading.Thread(target=worker, args=(2, "matern52")),
          threading.Thread(target=worker, args=(3, "netdiffusion"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


# ---- RCCL at world = 1, and the launcher ---------------------------------------------------------------------------------
_NCCL_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
from oracle import gp_oracle as O
import seaiceextentforecasting_amd as S
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))        # RCCL
assert dist.get_backend() == "nccl"
# year sharding: .cuda() all-gather of the per-fit results on RCCL
B, n, d, F = 3, 300, 4, 7
Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 1, d))
for b in range(B):
    Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 50 + b, m=1)
ell = 1.0 + 0.1 * np.arange(F); sn = 0.05 + 0.01 * np.arange(F)
with S.GPR(kernel="rbf") as gp:
    res = S.fit_batch_sharded(lambda Xl, yl, Xsl, e, s: gp.fit_batch(Xl, yl, Xsl, e, s, concurrency=1, group=4), Xb, yb, Xsb, ell, sn, rank, world, dist)
for i in range(F):
    r = O.fit_predict(Xb[i %% B], yb[i %% B], Xsb[i %% B], ell[i], sn[i], kind="rbf", ref_idiom=False)
    assert abs(res["nlml"][i] - r["nlml"]) <= 1e-9 * abs(r["nlml"]) and abs(res["mean"][i, 0] - r["fmean"][0]) <= 1e-8 * abs(r["fmean"][0]), i
# sharded Cholesky through the LIBRARY's RCCL communicator (sigp_dist_init with a unique id; one rank here, so every panel
# broadcast and both all-reduces are real RCCL calls on the library's streams).  torch is loaded: the library binds torch's RCCL.
X, y, Xs = O.synthetic_problem(1100, 8, 4242, m=3)
ref = O.fit_predict(X, y, Xs, np.sqrt(8.0), 1e-2, kind="rbf", ref_idiom=False)
relf = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))
for la in (True, False):
    with S.DistributedGPR("rbf", rank, world, dist, device=0, outer_blocks=2, lookahead=la, force_rccl=True, stats=True) as dg:
        assert dg.transport == "rccl"
        dg.fit(X, y, np.sqrt(8.0), 1e-2, Xs=Xs)
        mu, var = dg.predict(Xs)
        st = dg.stats()
        nl = dg.nlml_
    assert relf(mu, ref["fmean"]) <= 1e-8 and relf(var, ref["fvar"]) <= 1e-8 and relf(nl, ref["nlml"]) <= 1e-9
    assert st["collectives"] >= 4 + 2 and st["bcast_bytes"] > 0, st          # 5 panels (the last one has no reader and stays home) + the two all-reduces went through RCCL
Xf, yf, Xsf = O.synthetic_problem(900, 16, 515, m=2)
reff = O.fit_predict(Xf, yf, Xsf, 4.0, 1e-1, kind="matern52", ref_idiom=False)
with S.DistributedGPR("matern52", rank, world, dist, device=0, outer_blocks=2, dtype="f32", force_rccl=True) as dg:
    dg.fit(Xf, yf, 4.0, 1e-1, Xs=Xsf)
    mu, var = dg.predict(Xsf)
    resid = dg.refine_residual_
assert relf(mu, reff["fmean"]) <= 1e-6 and relf(var, reff["fvar"]) <= 1e-5 and 0 < resid <= 1e-10
# the collective itself: a device tensor through RCCL
t = torch.arange(8, dtype=torch.float64, device="cuda")
dist.broadcast(t, src=0); dist.all_reduce(t)
assert float(t.sum().item()) == 28.0 * world
dist.barrier(); dist.destroy_process_group()
open(os.path.join(%(out)r, "ok_%%d" %% rank), "w").write("ok")
'''


def test_rccl_backend_world1(tmp_path):
    """The nccl (= RCCL) code paths -- .cuda() gathers of fit_batch_sharded, the sharded fit on the library's own RCCL
    communicator (fp64 and fp32), init with device_id -- executed once on the box's single GPU (world = 1)."""
    script = tmp_path / "worker.py"
    script.write_text(_NCCL_WORKER % dict(root=ROOT, out=str(tmp_path)))
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert (tmp_path / "ok_0").exists()


def test_bench_gpus_flag_fails_loudly_without_enough_gpus():
    """`python bench.py --gpus N` on a box with fewer than N GPUs must not silently measure one GPU."""
    import torch
    ndev = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout)
    env = dict(os.environ); env["WORLD_SIZE"] = "1"; env["RANK"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_two_ranks_sharing_the_gpu_strong_scaling(tmp_path):
    """bench.py --gpus 2 --scaling strong spawns two ranks itself (gloo transport: both share the one GPU) and prints one
    line with n_gpus = 2; small n so it takes seconds."""
    env = dict(os.environ); env["SIGP_BENCH_BACKEND"] = "gloo"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--n", "1024",
                        "--years", "8", "--group", "8", "--no-cpu-baseline", "--no-extras", "--no-profile"], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    import json
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0


def test_networks_driver_on_the_device_matches_the_host_driver(S):
    """networks(dataset, engine=gp): tau and the area sums on the GPU, the areas from sigp_area_level -- same nodes and the same
    anomaly series (the GP's features) as the host driver on the realistic 57 x 57 grid."""
    import seaiceextentforecasting_amd.networks as NW
    rng = np.random.default_rng(11)
    base = rng.standard_normal((6, 41))
    field = np.full((57, 57, 41), np.nan)
    for i in range(57):
        for j in range(57):
            if (i - 28) ** 2 + (j - 28) ** 2 < 26 ** 2:
                field[i, j] = base[(i // 20) * 2 + (j // 30)] + 0.7 * rng.standard_normal(41)
    area = rng.uniform(0.5, 1.5, (57, 57))
    host = {"dt": field.copy(), "psar": area}
    NW.networks(host)
    with S.GPR(kernel="rbf") as gp:
        dev = {"dt": field.copy(), "psar": area}
        NW.networks(dev, engine=gp)
    assert list(host["nodes"]) == list(dev["nodes"]) and host["nodes"] == dev["nodes"]
    for k in host["anoms"]:
        assert np.array_equal(host["anoms"][k], dev["anoms"][k]), k


def test_retro_run_end_to_end_device_pipeline_equals_host_pipeline(S):
    """The retro scripts' main sequence (September1st_retro.py:295-299: detrend -> networks -> forecast -> skill) on synthetic
    inputs of the real shape (57 x 57 grid, three regions), once with detrending / tau / area sums on the GPU and once on the
    host (areas from sigp_area_level both times, the GP batch on the GPU both times): the same `.round(3)` forecasts and skills."""
    import seaiceextentforecasting_amd.networks as NW
    from scipy.stats import linregress
    fmin, fmax = 1992, 1996
    T = fmax - 1979 + 1
    rng = np.random.default_rng(2024)
    base = np.cumsum(rng.standard_normal((6, T)), axis=1) * 0.3 + rng.standard_normal((6, T))
    sic = np.full((57, 57, T), np.nan)
    for i in range(57):
        for j in range(57):
            if (i - 28) ** 2 + (j - 28) ** 2 < 26 ** 2:
                sic[i, j] = 0.5 + 0.1 * base[(i // 20) * 2 + (j // 30)] + 0.07 * rng.standard_normal(T) - 0.002 * np.arange(T)
    psar = rng.uniform(0.8, 1.2, (57, 57))
    regions = ["Pan-Arctic", "Beaufort", "Chukchi"]
    SIEs, SIEs_dt, SIEs_trend = {}, {}, {}
    for r, scale in zip(regions, (6.0, 0.6, 0.5)):
        SIEs[r] = (scale - 0.01 * scale * np.arange(T) + 0.05 * scale * (base[regions.index(r)] + rng.standard_normal(T))).round(3)
        trend = np.zeros((fmax - (fmin - 1) + 1, 2)); dt = np.zeros((fmax - (fmin - 1) + 1, T))
        for year in range(fmin - 1, fmax + 1):
            n = year - 1979 + 1
            reg = linregress(np.arange(n), SIEs[r][:n])
            trend[year - (fmin - 1)] = reg[0], reg[1]
            dt[year - (fmin - 1), :n] = SIEs[r][:n] - (reg[0] * np.arange(n) + reg[1])
        SIEs_trend[r], SIEs_dt[r] = trend, dt.round(3)
    outs = []
    with S.GPR(kernel="netdiffusion") as gp:
        for eng in (None, gp):
            SIC = {"data": sic.copy(), "psar": psar}
            S.detrend(SIC, fmin, fmax, engine=eng)
            NW.networks_retro(SIC, fmin, fmax, engine=eng)
            out = S.retro_forecast("north_September", SIC, SIEs_dt, SIEs_trend, fmin, fmax, gp=gp, batched=True)
            outs.append((out, S.skill(out, SIEs, SIEs_dt, fmin, fmax, regions), {k: v for k, v in SIC.items() if k.startswith("nodes_")}))
    assert outs[0][2] == outs[1][2]                                   # the same areas every year
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    assert outs[0][1][0] == outs[1][1][0] and outs[0][1][1] == outs[1][1][1]
