"""CPU tier 1: the oracle (oracle/gp_oracle.py) against captures of the reference's own forecast().

Bar: every intermediate of the GP block to <= 1e-12 relative (same LAPACK calls in the
same order, so in practice bit-equal or 1 ulp); MLII values incl. the inf branch; rounded retro outputs
exactly equal.
"""
import os

import numpy as np
import pytest

from oracle import gp_oracle as O


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def base_script(g):
    return g["script"].replace("_retro", "")


def run_oracle(g, ref_idiom):
    caps = []

    def fit(X, y, Xs, ell, sn, M):
        r = O.fit_predict(X, y, Xs, ell, sn, M=M, ref_idiom=ref_idiom)
        r.update(X=X, Xs=Xs, y=y)
        caps.append(r)
        return r

    if g["kind"] == "retro":
        out = O.retro_forecast(base_script(g), g["SIC"], g["SIEs_dt"], g["SIEs_trend"], g["args"][0], g["args"][1],
                               SST=g["SST"], fit=fit)
    else:
        SIC = {"anoms": g["SIC"]["anoms"]}
        SST = {"anoms": g["SST"]["anoms"]} if g["SST"] else None
        out = O.operational_forecast(base_script(g), SIC, g["SIEs_dt"], g["SIEs_trend"], g["args"][0], SST=SST, fit=fit)
    return out, caps


def test_reference_idiom_matches_reference_captures(golden):
    g = golden
    out, caps = run_oracle(g, ref_idiom=True)
    assert len(caps) == len(g["records"])
    for r, c in zip(g["records"], caps):
        assert c["X"].shape == r["X"].shape
        for ref_name, mine in [("X", "X"), ("Xs", "Xs"), ("y", "y"), ("M", "M"), ("Sigma_tilde", "Sigma_tilde"),
                               ("L_tilde", "L_tilde"), ("A_tilde", "A_tilde"), ("Sigma", "Sigma"), ("L", "L"),
                               ("alpha", "alpha"), ("KXXs", "KXXs"), ("v", "v")]:
            assert rel(c[mine], r[ref_name]) <= 1e-12, ref_name
        assert rel(c["sigma_f"], r["sigma_f"]) <= 1e-12
        assert rel(c["sigma_n"], r["sigma_n"]) <= 1e-12
        assert rel(c["kss"][0], r["KXsXs"][0][0]) <= 1e-12
        if g["kind"] == "op":
            assert rel(c["fmean"][0], r["fmean"]) <= 1e-12
            # fvar = k** - v^T v cancels: tolerance is relative to k**
            assert abs(c["fvar"][0] - r["fvar"]) <= 1e-13 * abs(float(r["KXsXs"][0][0]))
    if g["kind"] == "retro":
        for key, val in g["GPR"].items():
            np.testing.assert_array_equal(out[key], val, err_msg=key)     # .round(3) outputs: exact
    else:
        regs = O.SCRIPT_TABLE[base_script(g)]["regions"]
        for r in g["records"]:
            assert rel(out[regs[int(r["k"])]]["fmean_rt"], r["fmean_rt"]) <= 1e-12


def test_best_practice_path_agrees(golden):
    """ref_idiom=False (one Cholesky, triangular solves) agrees with the reference to rounding."""
    g = golden
    _, caps = run_oracle(g, ref_idiom=False)
    for r, c in zip(g["records"], caps):
        cond = np.linalg.cond(r["L_tilde"]) ** 2
        tol = max(1e-10, 50 * cond * 2.3e-16)
        assert rel(c["alpha"], r["alpha"]) <= tol
        assert rel(c["L"], r["L"]) <= 1e-12
        ref_mean = float((r["KXXs"].T @ r["alpha"])[0, 0])
        ref_var = float((r["KXsXs"] - r["v"].T @ r["v"])[0, 0])
        assert abs(c["fmean"][0] - ref_mean) <= tol * max(abs(ref_mean), np.abs(r["KXXs"]).max() * np.abs(r["alpha"]).max())
        assert abs(c["fvar"][0] - ref_var) <= tol * abs(float(r["KXsXs"][0, 0]))


def test_mlii_matches_reference_closure(golden):
    g = golden
    for r in g["records"]:
        X, y, M = r["X"], r["y"], r["M"]
        for th, nl, gr in zip(r["mlii_theta"], r["mlii_nlml"], r["mlii_grad"]):
            nl_o, gr_o = O.mlii(th, X, y, M=M, grad="ref")
            if np.isinf(nl):
                assert np.isinf(nl_o) and np.all(np.isinf(gr_o))
            else:
                assert rel(nl_o, nl) <= 1e-12
                # the trace-of-solve terms amplify rounding by cond(K); both sides run the same LAPACK calls
                assert rel(gr_o, gr) <= 1e-9
    assert any(np.isinf(r["mlii_nlml"]).any() for r in g["records"]), "inf branch must be covered"


def test_identities_of_the_reference_block(golden):
    """SURVEY App. A identities: y^T alpha = n, L = sqrt(sf) L~, alpha = A~/sf, M rows sum to 0."""
    for r in golden["records"]:
        n = r["y"].shape[0]
        assert abs(float((r["y"].T @ r["alpha"])[0, 0]) - n) <= 1e-8 * n
        assert rel(np.sqrt(r["sigma_f"]) * r["L_tilde"], r["L"]) <= 1e-12
        assert rel(r["A_tilde"] / r["sigma_f"], r["alpha"]) <= 1e-9
        assert np.abs(r["M"].sum(0)).max() <= 1e-12 * max(1.0, np.abs(r["M"]).max())


def test_exact_gradient_is_the_derivative_of_nlml():
    """grad='exact' (new mode) passes a central finite-difference check; grad='ref' does not (App. C-7)."""
    rng = np.random.default_rng(5)
    n, N = 30, 6
    X = rng.standard_normal((n, N))
    y = (X @ rng.standard_normal(N) + 0.3 * rng.standard_normal(n)).reshape(-1, 1)
    for kind, th in [("netdiffusion", np.array([np.log(0.14), np.log(6.1)])), ("rbf", np.array([0.5, -1.0])),
                     ("matern52", np.array([0.8, -0.5]))]:
        f0, g0 = O.mlii(th, X, y, kind=kind, grad="exact")
        h = 1e-5
        fd = np.array([(O.mlii(th + h * e, X, y, kind=kind, grad="exact")[0]
                        - O.mlii(th - h * e, X, y, kind=kind, grad="exact")[0]) / (2 * h) for e in np.eye(2)])
        assert np.allclose(g0, fd, rtol=1e-5, atol=1e-7), (kind, g0, fd)


def test_rbf_matern_against_sklearn():
    """RBF / Matern-5/2 are not in the reference ('parity unpinned' there): cross-check the kernel
    functions against scikit-learn (not a reference dependency) at n in {64, 257}."""
    sk = pytest.importorskip("sklearn.gaussian_process.kernels")
    for n, d, ell in [(64, 4, 2.0), (257, 8, np.sqrt(8.0))]:
        X, _, Xs = O.synthetic_problem(n, d, 11, m=5)
        assert rel(O.cov_unit("rbf", X, Xs, ell), sk.RBF(length_scale=ell)(X, Xs)) <= 1e-13
        assert rel(O.cov_unit("matern52", X, Xs, ell), sk.Matern(length_scale=ell, nu=2.5)(X, Xs)) <= 1e-13


def test_config1_n64_d4_rbf_cpu_path():
    """BASELINE.json configs[0]: n=64, d=4 synthetic RBF GPR on the CPU path (plumbing)."""
    X, y, Xs = O.synthetic_problem(64, 4, 20240000, m=3)
    a = O.fit_predict(X, y, Xs, 2.0, 1e-2, kind="rbf", ref_idiom=True)
    b = O.fit_predict(X, y, Xs, 2.0, 1e-2, kind="rbf", ref_idiom=False)
    assert rel(a["fmean"], b["fmean"]) <= 1e-9 and rel(a["fvar"], b["fvar"]) <= 1e-9
    assert abs(float(y @ a["alpha"][:, 0]) - 64) < 1e-8 * 64
    assert np.all(a["fvar"] > 0)


def test_lean_oracle_path_is_the_same_arithmetic():
    """oracle.fit_predict_lean (row-block build, blocked Cholesky and block substitution for sizes where LAPACK's potrf in this
    image's OpenBLAS segfaults) against oracle.fit_predict, and the committed configs[4] fixture's header."""
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))
    for kind in ("rbf", "matern52"):
        X, y, Xs = O.synthetic_problem(900, 8, 3, m=3)
        a = O.fit_predict(X, y, Xs, np.sqrt(8.0), 1e-2, kind=kind, ref_idiom=False)
        b = O.fit_predict_lean(X, y, Xs, np.sqrt(8.0), 1e-2, kind=kind, row_block=128, threads=4)
        assert rel(b["fmean"], a["fmean"]) <= 1e-11 and rel(b["fvar"], a["fvar"]) <= 1e-11
        assert rel(b["nlml"], a["nlml"]) <= 1e-13 and rel(b["sigma_f"], a["sigma_f"]) <= 1e-13 and rel(b["A_tilde"], a["A_tilde"]) <= 1e-11
    rng = np.random.default_rng(0)
    A = rng.standard_normal((1500, 40)); K = A @ A.T; K[np.arange(1500), np.arange(1500)] += 5.0
    assert rel(O._cholesky_blocked(K.copy(), bs=333), np.linalg.cholesky(K)) <= 1e-13
    b = rng.standard_normal((1500, 3)); L = np.linalg.cholesky(K)
    assert rel(O._tri_solve_blocked(L, b, bs=200), np.linalg.solve(L, b)) <= 1e-12
    assert rel(O._tri_solve_blocked(L, b, trans=True, bs=200), np.linalg.solve(L.T, b)) <= 1e-12
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "config4_oracle.npz"))
    assert int(z["n"]) == 32768 and int(z["d"]) == 32 and z["A_tilde"].shape == (32768,) and np.isfinite(float(z["nlml"]))
