"""CPU tier: the C-ABI library loads and exports every symbol include/sigp.h declares (no compute calls),
the product path fails loudly without a GPU, and the host logic (feature rules, tables, sharding)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import gp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from seaiceextentforecasting_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "seaiceextentforecasting_amd", "csrc")])
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "sigp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sigp_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = lib.load()
    for name in declared:
        assert hasattr(L, name), "libsigp.so does not export %s" % name
    assert declared == set(lib.SIGNATURES), "ctypes table and header disagree: %s" % (declared ^ set(lib.SIGNATURES))
    assert L.sigp_version() >= 100


def test_product_path_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from seaiceextentforecasting_amd import GPR
    with pytest.raises(lib.SigpError):
        GPR(kernel="rbf")


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "seaiceextentforecasting_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no CPU fallback", ""), fn


def test_feature_rules_match_the_oracle(golden):
    """features.select_features / design_matrix / laplacian_M (vectorised) pick exactly what the reference did."""
    import seaiceextentforecasting_amd as S
    g = golden
    script = g["script"].replace("_retro", "")
    tab = S.SCRIPT_TABLE[script]
    otab = O.SCRIPT_TABLE[script]
    assert np.allclose(tab["ell"], otab["ell"], rtol=0, atol=0) and np.allclose(tab["sn"], otab["sn"], rtol=0, atol=0)
    for r in g["records"]:
        k = int(r["k"])
        if g["kind"] == "retro":
            year = int(r["year"])
            key = "anoms_%d" % (year - 1 if tab["drop_first"] else year)
            sic, sst = g["SIC"][key], (g["SST"][key] if g["SST"] else None)
        else:
            sic, sst = g["SIC"]["anoms"], (g["SST"]["anoms"] if g["SST"] else None)
        feats = S.select_features(r["y"], sic, sst, rule=tab["rule"], k=k, pthr=tab["pthr"])
        X, Xs = S.design_matrix(feats, tab["standardise"])
        assert X.shape == r["X"].shape
        assert np.max(np.abs(X - r["X"])) <= 1e-13 * max(1.0, np.abs(r["X"]).max())
        assert np.max(np.abs(Xs - r["Xs"])) <= 1e-13 * max(1.0, np.abs(r["Xs"]).max())
        M = S.laplacian_M(X)
        assert np.array_equal(S.laplacian_M(r["X"]), r["M"])      # bit-identical on the reference's own X
        assert np.max(np.abs(M - r["M"])) <= 1e-12 * max(1.0, np.abs(r["M"]).max())
        assert X.flags["C_CONTIGUOUS"]


def test_pearson_vectorised_equals_scipy():
    from scipy.stats import pearsonr
    from seaiceextentforecasting_amd.features import pearson_rp
    rng = np.random.default_rng(0)
    y = rng.standard_normal(37)
    A = rng.standard_normal((20, 37)) + 0.3 * y
    r, p = pearson_rp(y, A)
    for i in range(20):
        r0, p0 = pearsonr(y, A[i])
        assert abs(r[i] - r0) <= 1e-14 and abs(p[i] - p0) <= 1e-12 * max(p0, 1e-300) + 1e-15


def test_empty_feature_set_raises_like_the_reference():
    import seaiceextentforecasting_amd as S
    with pytest.raises(IndexError):
        S.design_matrix([], False)


def test_shard_indices_cover_everything_once():
    import seaiceextentforecasting_amd as S
    for n_items in (0, 1, 7, 40, 41):
        for world in (1, 2, 3, 8):
            allidx = np.concatenate([S.shard_indices(n_items, r, world) for r in range(world)])
            assert sorted(allidx.tolist()) == list(range(n_items))
    with pytest.raises(ValueError):
        S.shard_indices(5, 3, 2)


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from oracle import gp_oracle as O
import seaiceextentforecasting_amd as S
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
B, n, d, F = 3, 40, 3, 7
Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 1, d))
for b in range(B):
    Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 50 + b, m=1)
ell = 1.0 + 0.1 * np.arange(F); sn = 0.05 + 0.01 * np.arange(F)
def engine(Xl, yl, Xsl, e, s):          # CPU stand-in for GPR.fit_batch (the GPU engine is exercised by -m gpu)
    out = dict(sigma_f=[], nlml=[], info=[], sigma_n=[], mean=[], var=[])
    for i in range(len(e)):
        r = O.fit_predict(Xl[i], yl[i], Xsl[i], e[i], s[i], kind="rbf", ref_idiom=False)
        out["sigma_f"].append(r["sigma_f"]); out["nlml"].append(r["nlml"]); out["info"].append(0)
        out["sigma_n"].append(r["sigma_n"]); out["mean"].append(r["fmean"]); out["var"].append(r["fvar"])
    return {k: np.asarray(v) for k, v in out.items()}
res = S.fit_batch_sharded(engine, Xb, yb, Xsb, ell, sn, rank, world, dist)
for i in range(F):
    r = O.fit_predict(Xb[i %% B], yb[i %% B], Xsb[i %% B], ell[i], sn[i], kind="rbf", ref_idiom=False)
    assert abs(res["nlml"][i] - r["nlml"]) < 1e-12 and abs(res["mean"][i, 0] - r["fmean"][0]) < 1e-12, (rank, i)
assert res["mean"].shape == (F, 1)
dist.barrier(); dist.destroy_process_group()
open(os.path.join(%(out)r, "ok_%%d" %% rank), "w").write("ok")
'''


def test_year_sharding_over_two_ranks_gloo(tmp_path):
    """N>1 path: fits dealt round-robin over ranks, results all-gathered (world_size 2, gloo, CPU)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT, out=str(tmp_path)))
    port = 29500 + (os.getpid() % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists(), p.stdout[-2000:] + p.stderr[-2000:]


def _callers_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "callers.npz"), allow_pickle=False)


@pytest.mark.parametrize("tag,regions", [("north", ["Pan-Arctic", "Beaufort", "Chukchi"]), ("south", ["Pan-Antarctic", "Ross", "Weddell"])])
def test_skill_matches_reference(tag, regions):
    """callers.skill == the reference's skill() (June1st_retro.py:293-314 / February1st_retro.py) on the golden inputs."""
    import seaiceextentforecasting_amd as S
    z = _callers_golden()
    fmin, fmax = [int(v) for v in z["skill_%s/args" % tag]]
    SIEs = {r: z["skill_%s/SIEs/%s" % (tag, r)] for r in regions}
    SIEs_dt = {r: z["skill_%s/SIEs_dt/%s" % (tag, r)] for r in regions}
    GPR = {r + k: z["skill_%s/GPR/%s%s" % (tag, r, k)] for r in regions for k in ("_fmean", "_fvar", "_fmean_rt")}
    srt, sdt, dto = S.skill(GPR, SIEs, SIEs_dt, fmin, fmax, regions)
    np.testing.assert_array_equal(np.array(srt, dtype=np.float64), z["skill_%s/skill_rt" % tag])
    np.testing.assert_array_equal(np.array(sdt, dtype=np.float64), z["skill_%s/skill_dt" % tag])
    np.testing.assert_array_equal(np.array(dto, dtype=np.float64), z["skill_%s/dt_obs" % tag])
    df_dt, df_rt = S.forecast_tables(GPR, SIEs, SIEs_dt, fmin, fmax, regions)
    assert list(df_dt.columns)[:3] == [regions[0] + "$_o$", regions[0] + "$_f$", regions[0] + "$_f$ unc"]
    assert df_dt.index[-1] == "Skill" and df_rt.shape == (fmax - fmin + 2, 6)
    assert df_dt.iloc[-1, 1] == srt[0] * 0 + sdt[0] and df_rt.iloc[-1, 1] == srt[0]
    assert df_dt.iloc[0, 2] == np.sqrt(GPR[regions[0] + "_fvar"]).round(3)[0]


def test_detrend_matches_reference():
    """Vectorised detrend == the reference's per-pixel linregress loop (north/June1st.py:179-194 and the retro form),
    including all-NaN pixels and NaN propagation."""
    import seaiceextentforecasting_amd as S
    z = _callers_golden()
    data = z["detrend_retro/data"]
    fmin, fmax = [int(v) for v in z["detrend_retro/args"]]
    ds = S.detrend({"data": data.copy()}, fmin, fmax)
    for year in range(fmin, fmax + 1):
        for key in ("dt_%d" % year, "trend_%d" % year):
            ref = z["detrend_retro/" + key]
            assert np.array_equal(np.isnan(ds[key]), np.isnan(ref)), key
            assert np.nanmax(np.abs(ds[key] - ref)) <= 1e-12 * max(1.0, np.nanmax(np.abs(ref))), key
    ds2 = S.detrend({"data": data.copy()})
    assert np.array_equal(np.isnan(ds2["dt"]), np.isnan(z["detrend_op/dt"]))
    assert np.nanmax(np.abs(ds2["dt"] - z["detrend_op/dt"])) <= 1e-12 and np.nanmax(np.abs(ds2["trend"] - z["detrend_op/trend"])) <= 1e-12


@pytest.mark.parametrize("native", [False, True])
@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_complex_networks_restatement_matches_reference(case, native):
    """networks.Network == ComplexNetworks.Network (reference module run on synthetic fields, goldens
    tests/golden/networks_*.npz): tau, the areas V (same ids, same cells in the same order), the anomaly series that
    become the GP's features, links, strength map.  area_level both as the Python restatement and as the C++ host
    function of libsigp.so (sigp_area_level)."""
    import seaiceextentforecasting_amd.networks as NW
    z = np.load(os.path.join(ROOT, "tests", "golden", "networks_%s.npz" % case), allow_pickle=False)
    data, aux, latlon = z["data"], z["aux"], bool(int(z["latlon"]))
    net = NW.Network(data=data.copy())
    NW.Network.tau(net, 0.01)
    assert net.tau == float(z["tau"])
    NW.Network.area_level(net, latlon_grid=latlon, native=native)
    ids = [int(i) for i in z["area_ids"]]
    assert list(net.V.keys()) == ids
    for k in ids:
        assert np.array_equal(np.array(net.V[k], dtype=np.int64), z["V/%d" % k]), k
    if latlon:
        NW.Network.intra_links(net, lat=aux)
    else:
        NW.Network.intra_links(net, area=aux)
    for k in ids:
        assert np.array_equal(net.anomaly[k], z["anomaly/%d" % k])
        assert np.allclose(net.links[k], z["links/%d" % k], rtol=1e-13, atol=1e-13)
        assert abs(net.strength[k] - float(z["strength/%d" % k])) <= 1e-12 * abs(float(z["strength/%d" % k]))
    assert np.array_equal(np.isnan(net.strengthmap), np.isnan(z["strengthmap"]))


def test_native_area_level_decides_like_the_python_restatement():
    """sigp_area_level (C++, host) against networks.Network.area_level(native=False) on random fields of several shapes, plain
    and lat-lon grids: same area ids in the same order, same cells in the same order, same `unavail` list -- which needs every
    mean to be bit-identical to np.nanmean (sigp_host_nanmean: NumPy's pairwise summation order, checked here on lengths
    around its 8 / 128 block boundaries, with NaNs, all-NaN and empty)."""
    import warnings
    import seaiceextentforecasting_amd.networks as NW
    from seaiceextentforecasting_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(0)
    for n in list(range(0, 20)) + [127, 128, 129, 135, 136, 137, 255, 256, 257, 511, 513, 1025, 2109]:
        for rep in range(3):
            a = rng.standard_normal(n)
            if n and rep == 1:
                a[rng.integers(0, n, max(1, n // 5))] = np.nan
            if rep == 2:
                a[:] = np.nan
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = np.nanmean(a)
            got = lib.sigp_host_nanmean(L.ptr(np.ascontiguousarray(a)), n)
            assert (np.isnan(ref) and np.isnan(got)) or ref == got, (n, rep, ref, got)

    def field(seed, dimX, dimY, T):
        r = np.random.default_rng(seed)
        base = r.standard_normal((6, T))
        f = np.full((dimX, dimY, T), np.nan)
        for i in range(dimX):
            for j in range(dimY):
                if (i - dimX / 2) ** 2 / (0.45 * dimX) ** 2 + (j - dimY / 2) ** 2 / (0.45 * dimY) ** 2 < 1 and r.random() > 0.05:
                    f[i, j] = base[((i * 3 // dimX) * 2 + (j * 2 // dimY)) % 6] + r.uniform(0.4, 1.5) * r.standard_normal(T)
        return f
    for seed in range(8):
        f = field(seed, 14 + seed % 4 * 3, 12 + seed % 3 * 4, 30 + seed % 4 * 5)
        for latlon in (False, True):
            a = NW.Network(data=f.copy()); NW.Network.tau(a, 0.01)
            b = NW.Network(data=f.copy()); NW.Network.tau(b, 0.01)
            ea = eb = None
            try:
                NW.Network.area_level(a, latlon_grid=latlon, native=False)
            except ValueError as e:
                ea = e
            try:
                NW.Network.area_level(b, latlon_grid=latlon, native=True)
            except ValueError as e:
                eb = e
            assert (ea is None) == (eb is None), (seed, latlon)
            if ea is None:
                assert list(a.V) == list(b.V) and a.V == b.V and a.unavail == b.unavail, (seed, latlon)


def test_networks_driver_feeds_the_feature_rules():
    """networks() -> dataset['anoms'] has the contract features.select_features consumes (dict id -> series of length T)."""
    import seaiceextentforecasting_amd as S
    import seaiceextentforecasting_amd.networks as NW
    z = np.load(os.path.join(ROOT, "tests", "golden", "networks_a.npz"), allow_pickle=False)
    ds = {"dt": z["data"].copy(), "psar": z["aux"]}
    NW.networks(ds, latlon=False)
    T = z["data"].shape[2]
    assert all(v.shape == (T,) for v in ds["anoms"].values())
    y = np.random.default_rng(0).standard_normal((T - 1, 1))
    feats = S.select_features(y, ds["anoms"], rule="all_then_pos_p", k=0, pthr=0.05)
    X, Xs = S.design_matrix(feats, False)
    assert X.shape == (T - 1, len(ds["anoms"])) and Xs.shape == (1, len(ds["anoms"]))


def test_sigma_eigh_matches_expm_on_golden_laplacians():
    """one eigendecomposition per data set reproduces expm(l M) and M expm(l M) for the whole l grid,
    and the captured Sigma~ of the reference's own runs."""
    from scipy.linalg import expm
    from seaiceextentforecasting_amd.features import SigmaEigh, LGRID
    from conftest import load_golden, GOLDEN_NAMES
    checked = 0
    for name in GOLDEN_NAMES[:6]:
        for rec in load_golden(name)["records"][:3]:
            M = rec["M"]
            eig = SigmaEigh(M)
            for ell in list(LGRID[::4]) + [float(rec["ell"])]:
                if ell * np.abs(M).max() > 50:          # expm itself loses digits beyond this
                    continue
                ref = expm(ell * M)
                assert np.max(np.abs(eig.sigma(ell) - ref)) <= 1e-11 * np.max(np.abs(ref))
                assert np.max(np.abs(eig.msigma(ell) - M @ ref)) <= 1e-11 * np.max(np.abs(M)) * np.max(np.abs(ref))   # scale of the products summed
                if ell == float(rec["ell"]):
                    assert np.max(np.abs(eig.sigma(ell) - rec["Sigma_tilde"])) <= 1e-11 * np.max(np.abs(rec["Sigma_tilde"]))
                checked += 1
            S = eig.sigma(1e9)                           # extreme l: still a symmetric stochastic matrix (M has zero row sums)
            assert np.allclose(S, S.T) and np.allclose(S.sum(axis=1), 1.0, atol=1e-6)
    assert checked >= 10


def test_small_batch_factored_covariance_reproduces_expm():
    """smallbatch.SmallBatch host logic (no GPU): the factored form staged for the device, A diag(w) A^T, equals
    X expm(l M) X^T (north/June1st.py:264-265) in both modes -- eigh of M (weights exp(l lam)) and eigh of the Pade
    Sigma~ (weights = its eigenvalues) -- and the test rows give k*, k**."""
    from scipy.linalg import expm
    from seaiceextentforecasting_amd.smallbatch import SmallBatch
    from seaiceextentforecasting_amd.features import laplacian_M

    class FakeEngine:
        kernel, dtype = "netdiffusion", "f64"

    rng = np.random.default_rng(0)
    X = rng.standard_normal((30, 12)); y = rng.standard_normal(30); Xs = rng.standard_normal((2, 12))
    M = laplacian_M(X)
    sb = SmallBatch(FakeEngine())
    ds = sb.add_dataset(X, y, Xs, M)
    for ell in (1e-3, 0.05, 2.0):
        Sig = expm(ell * M)
        for mode in ("eigh", "pade"):
            sb.add_fit(ds, ell, 0.1, expm=mode)
            A, yy, lam, lam_mode, dlam = sb._sets[sb._fits[-1][0]]
            w = np.exp(ell * lam) if lam_mode == 0 else np.maximum(lam, 0.0)
            XX = np.vstack([X, Xs])
            full = (A * w) @ A.T
            want = XX @ Sig @ XX.T
            assert np.max(np.abs(full - want)) <= 1e-12 * np.max(np.abs(want)), (ell, mode)
            # the MLII gradient's X (M Sigma~) X^T (north/June1st.py:248) is a reweighting of the same A
            dw = lam * w if lam_mode == 0 else dlam
            dwant = XX @ (M @ Sig) @ XX.T
            assert np.max(np.abs((A * dw) @ A.T - dwant)) <= 1e-10 * max(np.max(np.abs(dwant)), np.max(np.abs(want))), (ell, mode)
            assert A.shape == (32, 12) and np.array_equal(yy, y)
    assert len(sb._sets) == 1 + 3            # one shared eigh set + one Pade set per l
    with pytest.raises(ValueError):
        sb.add_dataset(np.zeros((200, 3)), np.zeros(200))
    with pytest.raises(ValueError):
        sb.add_fit(ds, -1.0, 0.1)


def test_bench_refuses_to_run_a_multi_gpu_job_on_one_process():
    """`--gpus N` must never silently measure one GPU.  Without a GPU the launcher exits with a message; with
    a torchrun environment that disagrees with --gpus it exits too."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=200, env=env)
    assert p.returncode != 0 and ("needs a GPU" in p.stderr or "GPU(s) visible" in p.stderr), p.stderr[-500:]
    env["WORLD_SIZE"] = "4"; env["RANK"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=200, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr, p.stderr[-500:]


def test_bench_grid_and_strong_scaling_bookkeeping():
    """bench.py's step -> grid point map covers the SURVEY 8(d) ranges, and the strong-scaling deal covers every fit once."""
    sys.path.insert(0, ROOT)
    import bench
    ells, sns = bench.grid_axes(8, "full")
    assert len(ells) == 20 and len(sns) == 20 and np.isclose(ells[0], np.sqrt(8) * 0.1) and np.isclose(ells[-1], np.sqrt(8) * 10)
    assert np.isclose(sns[0], 1e-3) and np.isclose(sns[-1], 10.0)
    pts = {bench.grid_point(i, 8, "smoke") for i in range(16)}
    assert len(pts) == 16 and bench.grid_point(16, 8, "smoke") == bench.grid_point(0, 8, "smoke")
    G, years, steps = 40, 40, 3
    for world in (1, 2, 4, 8):
        seen = []
        for rank in range(world):
            mine = np.arange(steps * G)[rank::world]
            yr = mine % years
            period = len(np.unique(yr))
            assert np.array_equal(yr, np.tile(yr[:period], len(yr) // period + 1)[:len(yr)])     # what run_batch's (first + i) % sets assumes
            seen += list(mine)
        assert sorted(seen) == list(range(steps * G))


def test_asan_build_of_the_shim_rejects_bad_arguments_cleanly():
    """SURVEY 5 (sanitizers): host-side AddressSanitizer build of the C shim (make asan; GPU ASan is unavailable on the pool) --
    every entry point called with a null handle / bad sizes under ASan: clean rejections, no sanitizer report."""
    csrc = os.path.join(ROOT, "seaiceextentforecasting_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL)
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no ASan runtime in this image")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_negative_paths.py")], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0 and "asan negative paths ok" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
    assert "AddressSanitizer" not in p.stderr


def _build_c_caller(tmp_path):
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "seaiceextentforecasting_amd")
    exe = str(tmp_path / "c_caller")
    subprocess.run(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_caller.c"), "-o", exe, "-L" + libdir, "-lsigp", "-lm", "-Wl,-rpath," + libdir,
                    "-Wl,--allow-shlib-undefined"], check=True, capture_output=True, text=True)
    return exe


def test_plain_c99_caller_compiles_and_links_against_the_abi(lib, tmp_path):
    """include/sigp.h is a C header (not only a C++ one) and libsigp.so links into a program with no Python and no C++ in it:
    examples/c_caller.c (single fit + the sharded fit with the unique id passed between forked ranks) builds with
    -std=c99 -pedantic -Werror and starts (usage message; nothing touches a GPU)."""
    import subprocess
    exe = _build_c_caller(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr


@pytest.mark.gpu
def test_plain_c_program_runs_the_hot_path_and_the_sharded_fit(tmp_path):
    """The drop-in boundary used from C, no Python in the process (examples/c_caller.c): north/June1st.py:264-277 through
    sigp_fit_predict and, call by call, sigp_kernel_build / sigp_potrf / sigp_fit / sigp_predict_ride (the program itself checks
    that the two agree bit for bit), predictions at new points, fp64 and fp32; then the sharded fit through the library's own RCCL
    communicator (sigp_dist_unique_id -> sigp_dist_init -> sigp_dist_fit -> sigp_dist_predict -> sigp_dist_shutdown) with one rank
    on this one-GPU box.  All against the oracle."""
    exe = _build_c_caller(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cases = [("single", 700, 6, 3, "rbf", "f64"), ("single", 1300, 9, 2, "matern52", "f32"), ("sharded", 1500, 8, 2, "rbf", "f64"),
             ("sharded", 1100, 5, 1, "matern52", "f32")]
    for mode, n, d, m, kind, dtype in cases:
        X, y, Xall = O.synthetic_problem(n, d, 31 + n, m=2 * m)
        Xs, Xnew = Xall[:m], Xall[m:]
        ell, sn = float(np.sqrt(d)), 0.05
        path = str(tmp_path / "in.bin")
        np.concatenate([X.ravel(), y.ravel(), Xs.ravel(), Xnew.ravel()]).astype(np.float64).tofile(path)
        kid, dt = {"rbf": "1", "matern52": "2"}[kind], {"f64": "0", "f32": "1"}[dtype]
        args = [exe, mode] + (["1"] if mode == "sharded" else []) + [str(n), str(d), str(m), kid, dt, repr(ell), repr(sn)]
        args += (["3"] if mode == "sharded" else []) + [path]
        r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, (mode, kind, dtype, r.stdout[-500:], r.stderr[-2000:])
        def num(t):
            try:
                return float(t)
            except ValueError:
                return None
        lines = [ln.strip() for ln in r.stdout.splitlines()]           # (RCCL prints a version banner on stdout)
        v = np.array([num(ln) for ln in lines if num(ln) is not None and len(ln.split()) == 1])
        assert v.size == 4 + 4 * m, r.stdout
        ref = O.fit_predict(X, y, Xall, ell, sn, kind=kind, ref_idiom=False)
        tol = 1e-8 if dtype == "f64" else 1e-5
        rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(np.abs(np.asarray(b)), 1e-300)))
        assert rel(v[0], ref["sigma_f"]) <= tol and rel(v[1], ref["nlml"]) <= max(tol, 1e-9) * (50 if dtype == "f32" else 1) and v[2] == 0
        assert rel(v[3], ref["sigma_f"] * sn) <= tol
        assert rel(v[4:4 + m], ref["fmean"][:m]) <= tol and rel(v[4 + m:4 + 2 * m], ref["fvar"][:m]) <= tol
        assert rel(v[4 + 2 * m:4 + 3 * m], ref["fmean"][m:]) <= tol and rel(v[4 + 3 * m:], ref["fvar"][m:]) <= tol


def test_committed_pmc_summary_belongs_to_the_committed_kernel_code():
    """bench.py quotes roofline.traffic from profiles/r05_pmc_syrk128.json only while the sha of the kernel sources recorded in it equals the
    tree's (a stale file is refused, and the line then carries traffic = null).  The file committed with the tree must be the tree's."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    pm = json.load(open(os.path.join(root, "profiles", "r05_pmc_syrk128.json")))
    assert pm["kernel_code_sha16"] == bench.kernel_code_sha16()
    assert pm["traffic_bytes_per_launch"] > 0 and 0.5 < pm["mfma_pipe_busy_fraction"] <= 1.0
    for other in ("r05_pmc_kbuild.json", "r05_pmc_syrk128_f32.json"):        # the round's other PMC summaries exist and carry what DESIGN.md quotes from them
        d = json.load(open(os.path.join(root, "profiles", other)))
        assert d.get("hbm_total_GBps", d.get("tflops_in_kernel", 0)) > 0, other
    ch = json.load(open(os.path.join(root, "profiles", "r05_pmc_chain_kernels.json")))        # the panel stream's kernels, same passes, same code
    assert ch["kernel_code_sha16"] == bench.kernel_code_sha16()
    for k in ("diag_update_kernel<double", "panel_strip_kernel<double", "chain_link_kernel<double"):
        assert ch["kernels"][k]["rocprofv3_stats_avg_ms"] > 0 and 0 <= ch["kernels"][k]["mfma_pipe_busy_fraction"] <= 1.0, k


def test_bench_compact_line_fits_the_drivers_tail():
    """bench.py prints ONE line on stdout: the compact metric line (the driver keeps a 2 000-character tail).  Built from the committed full record
    it must stay below 1.5 KB and carry the contract's fields, `roofline`, `cpu_baseline`, the parity of the timed step and the other configs."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    for rnd in ("r04", "r05"):
        full = json.load(open(os.path.join(root, "profiles", "%s_bench_line_verbose.json" % rnd)))
        c = bench.compact_line(full)
        txt = json.dumps(c)
        assert len(txt) <= 1600, (rnd, len(txt))        # (the driver's tail is 2 000 characters)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["vs_baseline"] is None and c["dtype"] == "f64" and "workload" in c["config"] and "model" not in c["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    for k in ("c1_ms", "c3_ms", "c4_ms", "c4_frac", "mlii_g40_ms"):
        assert k in c and c[k] > 0, k
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-3


def test_lockstep_optimiser_drivers_on_the_oracle():
    """optim.py host logic (no GPU): both lockstep drivers, fed by the oracle's MLII (value + exact gradient) on the six (region, year)
    data sets of a golden retro run, reach stationary points at or below scipy's L-BFGS-B from the same x0 (north/June1st.py:259-262)."""
    from scipy.optimize import minimize
    from seaiceextentforecasting_amd.optim import bfgs_lockstep, newton_lockstep
    from seaiceextentforecasting_amd.retro import _problem, _retro_inputs
    from seaiceextentforecasting_amd.features import SCRIPT_TABLE
    from conftest import load_golden
    g = load_golden("north_June_retro")
    fmin, fmax = g["args"]
    tab = SCRIPT_TABLE["north_June"]
    sets, x0 = [], []
    for k, region in enumerate(tab["regions"]):
        for year in range(fmin, fmax + 1):
            _, y, sic, sst = _retro_inputs(tab, g["SIC"], g["SIEs_dt"], g["SST"], region, year, fmin)
            X, Xs, M = _problem(tab, k, y, sic, sst)
            sets.append((X, y, M)); x0.append([np.log(tab["ell"][k]), np.log(tab["sn"][k])])
    x0 = np.array(x0)
    calls = {"n": 0}

    def ev(th, own=None):
        calls["n"] += 1
        own = range(len(th)) if own is None else own
        r = [O.mlii(t, sets[o][0], sets[o][1], M=sets[o][2], grad="exact") for t, o in zip(th, own)]
        return np.array([a for a, _ in r]), np.array([b for _, b in r])

    rn = newton_lockstep(ev, x0)
    assert rn["nfev"] == calls["n"] <= 12 and np.all(rn["converged"])
    calls["n"] = 0
    rb = bfgs_lockstep(ev, x0, maxiter=40)
    assert rb["nfev"] == calls["n"] <= 25 and np.all(rb["converged"])
    for i, (X, y, M) in enumerate(sets):
        sc = minimize(lambda t: O.mlii(t, X, y, M=M, grad="exact"), x0[i], jac=True, method="L-BFGS-B")
        for r in (rn, rb):
            assert r["fun"][i] <= sc.fun + 1e-6 * max(1.0, abs(sc.fun)) and np.max(np.abs(r["jac"][i])) <= 1e-4


def test_diagonal_block_role_map_is_a_permutation_for_every_simd_placement(tmp_path):
    """potrf_diag.hpp deals the eight waves' roles by the SIMD each wave sits on (read from HW_ID at run time).  Whatever the placement --
    all 4^8 tables -- the map must be a permutation of the roles: compiled for the host from the header's own text and brute-forced."""
    import re
    hdr = open(os.path.join(ROOT, "seaiceextentforecasting_amd", "csrc", "potrf_diag.hpp")).read()
    m = re.search(r"__device__ inline int diag_logical_wave\(.*?\n}\n", hdr, flags=re.S)
    assert m, "diag_logical_wave not found"
    body = m.group(0).replace("const volatile int* tab", "const int* tab")
    src = tmp_path / "roles.cpp"
    src.write_text('#include <cstdio>\n#define __device__\nstatic inline int __builtin_amdgcn_readfirstlane(int x) { return x; }\n' + body + r'''
int main() {
  long bad = 0;
  for (int code = 0; code < 65536; ++code) {
    int tab[8];
    for (int w = 0; w < 8; ++w) tab[w] = (code >> (2 * w)) & 3;
    unsigned seen = 0;
    for (int pw = 0; pw < 8; ++pw) { const int r = diag_logical_wave(tab, pw); if (r >= 0 && r < 8) seen |= 1u << r; }
    bad += seen != 0xffu;
  }
  int usual[8] = {0, 2, 1, 3, 0, 2, 1, 3};       // observed placement: the pivot wave's SIMD mate is the store wave, (1, 6), (2, 3), (4, 5) are SIMD pairs
  const bool ok = diag_logical_wave(usual, 0) == 0 && diag_logical_wave(usual, 4) == 7 && diag_logical_wave(usual, 1) == 1 && diag_logical_wave(usual, 5) == 6
                  && diag_logical_wave(usual, 2) == 2 && diag_logical_wave(usual, 6) == 3;
  std::printf("%ld %d\n", bad, (int)ok);
  return 0;
}
''')
    exe = tmp_path / "roles"
    subprocess.check_call(["g++", "-O1", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out == ["0", "1"], out


def test_bench_line_guardian_prints_exactly_one_line_whatever_happens_to_rank0():
    """N > 1: rank 0 hands the metric line to a forked child before it enters the (never multi-GPU-run) sharded fits.  The child prints
    the LAST line it was given: the final one on a normal end, the provisional one when rank 0 is killed in between, nothing when
    there was none."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = "import os, sys, signal; sys.path.insert(0, %r); import bench; bench.start_line_guardian(); " % root
    cases = {
        "bench.guardian_hand_over('PROVISIONAL', final=False); bench.guardian_hand_over('FINAL', final=True)": "FINAL\n",
        "bench.guardian_hand_over('PROVISIONAL', final=False); os.kill(os.getpid(), signal.SIGKILL)": "PROVISIONAL\n",
        "bench.guardian_hand_over('PROVISIONAL', final=False); os._exit(3)": "PROVISIONAL\n",
        "os._exit(0)": "",
    }
    for body, want in cases.items():
        p = subprocess.run([sys.executable, "-c", head + body], capture_output=True, text=True, timeout=120)
        assert p.stdout == want, (body, p.stdout, p.stderr[-300:])
    # without a guardian (N = 1) the final line goes straight to stdout and a provisional one nowhere
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; bench.guardian_hand_over('P', final=False); "
                        "bench.guardian_hand_over('F', final=True)" % root], capture_output=True, text=True, timeout=120)
    assert p.stdout == "F\n"
