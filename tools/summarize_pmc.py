"""Summarise the rocprofv3 --pmc passes of tools/collect_profiles.sh into profiles/<round>_pmc_syrk128.json."""
import json, os, sys
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
base = "gpurun_out/%s/" % rnd
out = {"command": "rocprofv3 --pmc <C> --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile  (one pass per counter set; default lockstep group)",
       "kernel": "sigp::syrk128_kernel<double, false, false>", "notes": [],
       "kernel_code_sha16": bench.kernel_code_sha16()}      # bench.py quotes roofline.traffic from this file only while the kernel code is the same
try:
    G = int(json.load(open(base + "bench.json"))["config"]["fits_per_step"])      # lockstep members per launch of the profiled bench command
except Exception:
    G = 160
out["lockstep_group"] = G
sel = lambda df: df[df["Kernel_Name"].str.contains("syrk128_kernel<double, false", regex=False)]   # the trailing / inner updates (not the SET = true panel solve)
fe = sel(pd.read_csv(base + "pmc_FETCH_SIZE/bench_counter_collection.csv"))
wr = sel(pd.read_csv(base + "pmc_WRITE_SIZE/bench_counter_collection.csv"))
mf = sel(pd.read_csv(base + "pmc_mfma/bench_counter_collection.csv"))
out["launches"] = int(len(fe))
out["FETCH_SIZE_KB_per_launch_raw"] = float(fe["Counter_Value"].mean())
out["WRITE_SIZE_KB_per_launch_raw"] = float(wr["Counter_Value"].mean())
out["fetch_bytes_per_launch_corrected"] = float(fe["Counter_Value"].mean()) * 1024 * 2
out["write_bytes_per_launch"] = float(wr["Counter_Value"].mean()) * 1024
out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch_corrected"] + out["write_bytes_per_launch"]
g = mf.groupby("Counter_Name")["Counter_Value"].sum()
out["SQ_VALU_MFMA_BUSY_CYCLES_sum"] = float(g["SQ_VALU_MFMA_BUSY_CYCLES"])
out["GRBM_GUI_ACTIVE_sum_over_8_XCD"] = float(g["GRBM_GUI_ACTIVE"])
out["mfma_pipe_busy_fraction"] = float(g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["GRBM_GUI_ACTIVE"] / 8 * 1024))
out["notes"] += ["FETCH_SIZE / WRITE_SIZE are KB (x1024); gfx950: FETCH_SIZE reports half of a wide coalesced read stream (x2) -- MI355X_MICROARCH.md, HBM section",
                 "mfma_pipe_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs): share of SIMD-cycles with the matrix pipe busy, at the clock the chip actually held",
                 "algorithmic C traffic of the same launches (bench.py accounting): one read + one write of every 128x128 fp64 tile; WRITE_SIZE matches it to <1 %; FETCH also contains the L2-missing part of the A/B panel reads (served by the Infinity Cache)"]
json.dump(out, open("profiles/%s_pmc_syrk128.json" % rnd, "w"), indent=1)
print(out["launches"], "fetch GB %.3f write GB %.3f busy %.3f" % (out["fetch_bytes_per_launch_corrected"] / 1e9, out["write_bytes_per_launch"] / 1e9, out["mfma_pipe_busy_fraction"]))

# the covariance build (north_star: "rocprof achieved-HBM-GB/s on the kernel build"): PMC bytes of the same passes / rocprofv3 --stats duration
selk = lambda df: df[df["Kernel_Name"].str.contains("kbuild_kernel<double", regex=False)]
fk = selk(pd.read_csv(base + "pmc_FETCH_SIZE/bench_counter_collection.csv"))
wk = selk(pd.read_csv(base + "pmc_WRITE_SIZE/bench_counter_collection.csv"))
st = pd.read_csv(base + "stats/bench_kernel_stats.csv")
st = st[st["Name"].str.contains("kbuild_kernel<double", regex=False)]
kb = {"kernel": "sigp::kbuild_kernel<double, 8> (RBF covariance build of %d lockstep members, lower 64x128 tiles, n = 8192, d = 8)" % G,
      "launches_pmc": int(len(wk)), "write_bytes_per_launch": float(wk["Counter_Value"].mean()) * 1024,
      "fetch_bytes_per_launch_corrected": float(fk["Counter_Value"].mean()) * 1024 * 2,
      "algorithmic_bytes_per_launch": G * (4.0 * 8192 * 8193 + 8.0 * 8192 * 8),
      "rocprofv3_stats_avg_ms": float(st["AverageNs"].iloc[0]) / 1e6, "rocprofv3_stats_calls": int(st["Calls"].iloc[0])}
kb["hbm_write_GBps"] = kb["write_bytes_per_launch"] / (kb["rocprofv3_stats_avg_ms"] * 1e-3) / 1e9
kb["hbm_total_GBps"] = (kb["write_bytes_per_launch"] + kb["fetch_bytes_per_launch_corrected"]) / (kb["rocprofv3_stats_avg_ms"] * 1e-3) / 1e9
kb["frac_of_8TBps"] = kb["hbm_total_GBps"] / 8000.0
json.dump(kb, open("profiles/%s_pmc_kbuild.json" % rnd, "w"), indent=1)
print("kbuild: write GB %.2f fetch GB %.2f in %.3f ms -> %.2f TB/s" % (kb["write_bytes_per_launch"] / 1e9, kb["fetch_bytes_per_launch_corrected"] / 1e9, kb["rocprofv3_stats_avg_ms"], kb["hbm_total_GBps"] / 1e3))

# the chain kernels of the same bench command: what the panel stream's launches do with the chip while the trailing update shares it
try:
    fe_all = pd.read_csv(base + "pmc_FETCH_SIZE/bench_counter_collection.csv"); wr_all = pd.read_csv(base + "pmc_WRITE_SIZE/bench_counter_collection.csv")
    mf_all = pd.read_csv(base + "pmc_mfma/bench_counter_collection.csv")
    st_all = pd.read_csv(base + "stats/bench_kernel_stats.csv")
    try:
        lds_all = pd.read_csv(base + "pmc_lds/bench_counter_collection.csv")
    except Exception:
        lds_all = None
    chain = {"command": out["command"], "lockstep_group": G, "kernel_code_sha16": out["kernel_code_sha16"], "kernels": {},
             "notes": ["per launch, averaged over the launches of one bench command (PMC collection serialises kernels: these are the kernels ALONE on the chip, not beside the trailing update)",
                       "flops from the engine's own accounting are not in the PMC passes; mfma_pipe_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
                       "lds_bank_conflict_fraction = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles); waves_per_launch = SQ_WAVES"]}
    for key in ("diag_update_kernel<double", "panel_strip_kernel<double", "chain_link_kernel<double", "potrf_diag_kernel<double", "syrk128_kernel<double, true"):
        sel2 = lambda df, col: df[df[col].str.contains(key, regex=False)]
        fk2, wk2, mk2, sk2 = sel2(fe_all, "Kernel_Name"), sel2(wr_all, "Kernel_Name"), sel2(mf_all, "Kernel_Name"), sel2(st_all, "Name")
        if not len(fk2) or not len(sk2):
            continue
        gm = mk2.groupby("Counter_Name")["Counter_Value"].sum()
        e = {"launches_pmc": int(len(fk2)), "rocprofv3_stats_calls": int(sk2["Calls"].iloc[0]), "rocprofv3_stats_avg_ms": float(sk2["AverageNs"].iloc[0]) / 1e6,
             "rocprofv3_stats_total_ms": float(sk2["TotalDurationNs"].iloc[0]) / 1e6,
             "fetch_bytes_per_launch_corrected": float(fk2["Counter_Value"].mean()) * 1024 * 2, "write_bytes_per_launch": float(wk2["Counter_Value"].mean()) * 1024,
             "mfma_pipe_busy_fraction": float(gm["SQ_VALU_MFMA_BUSY_CYCLES"] / (gm["GRBM_GUI_ACTIVE"] / 8 * 1024)) if "GRBM_GUI_ACTIVE" in gm and gm["GRBM_GUI_ACTIVE"] > 0 else None}
        if "SQ_BUSY_CU_CYCLES" in gm and "GRBM_GUI_ACTIVE" in gm and gm["GRBM_GUI_ACTIVE"] > 0:
            e["cu_busy_fraction"] = float(gm["SQ_BUSY_CU_CYCLES"] / (gm["GRBM_GUI_ACTIVE"] / 8 * 256))
        if lds_all is not None:
            gl = sel2(lds_all, "Kernel_Name").groupby("Counter_Name")["Counter_Value"].sum()
            if "SQ_LDS_BANK_CONFLICT" in gl and "SQ_LDS_IDX_ACTIVE" in gl and gl["SQ_LDS_IDX_ACTIVE"] > 0:
                e["lds_bank_conflict_fraction"] = float(gl["SQ_LDS_BANK_CONFLICT"] / gl["SQ_LDS_IDX_ACTIVE"])
            lw = sel2(lds_all, "Kernel_Name")
            lw = lw[lw["Counter_Name"] == "SQ_WAVES"]
            if len(lw):
                e["waves_per_launch"] = float(lw["Counter_Value"].mean())
        e["hbm_total_GBps_alone"] = (e["fetch_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]) / (e["rocprofv3_stats_avg_ms"] * 1e-3) / 1e9
        chain["kernels"][key] = e
    json.dump(chain, open("profiles/%s_pmc_chain_kernels.json" % rnd, "w"), indent=1)
    print("chain kernels:", {k: (round(v["rocprofv3_stats_avg_ms"], 3), v["mfma_pipe_busy_fraction"]) for k, v in chain["kernels"].items()})
except FileNotFoundError as e:
    print("no chain-kernel passes:", e)

# the fp32 trailing update at configs[4]'s shape (one fit, n = 32768, d = 32): tools/collect_profiles.sh <round> f32
try:
    self = lambda df: df[df["Kernel_Name"].str.contains("syrk128_kernel<float, false", regex=False)]
    ff = self(pd.read_csv(base + "f32_pmc_FETCH_SIZE/f32_counter_collection.csv"))
    fw = self(pd.read_csv(base + "f32_pmc_WRITE_SIZE/f32_counter_collection.csv"))
    fs = pd.read_csv(base + "f32_stats/f32_kernel_stats.csv")
    fs = fs[fs["Name"].str.contains("syrk128_kernel<float, false", regex=False)]
    n = 32768
    f32 = {"kernel": "sigp::syrk128_kernel<float, false, false> (trailing + in-panel updates of one fp32 Cholesky, n = 32768, outer panels of 16 x 128: K = 2048)",
           "command": "rocprofv3 --pmc <C> / --kernel-trace --stats -- python3 tools/shard_profile.py --single --dtype f32 --n 32768 --d 32 --kernel matern52 --sn 0.1 --reps 2",
           "launches_per_run": int(len(ff)), "fetch_bytes_total_corrected": float(ff["Counter_Value"].sum()) * 1024 * 2, "write_bytes_total": float(fw["Counter_Value"].sum()) * 1024,
           "stats_total_ms": float(fs["TotalDurationNs"].iloc[0]) / 1e6, "stats_calls": int(fs["Calls"].iloc[0]), "fits_in_run": 3,
           "algorithmic_flops_per_fit": n ** 3 / 3.0}
    f32["tflops_in_kernel"] = f32["algorithmic_flops_per_fit"] * f32["fits_in_run"] / (f32["stats_total_ms"] * 1e-3) / 1e12
    f32["frac_of_fp32_mfma_peak"] = f32["tflops_in_kernel"] / 157.3
    try:
        fm = self(pd.read_csv(base + "f32_pmc_mfma/f32_counter_collection.csv")).groupby("Counter_Name")["Counter_Value"].sum()
        f32["mfma_pipe_busy_fraction"] = float(fm["SQ_VALU_MFMA_BUSY_CYCLES"] / (fm["GRBM_GUI_ACTIVE"] / 8 * 1024))
    except Exception:
        pass
    # the other HBM-side kernels of the same run: the refinement's one-pass residual and the fp32 build (bytes per launch from the same passes)
    fe_all = pd.read_csv(base + "f32_pmc_FETCH_SIZE/f32_counter_collection.csv"); wr_all = pd.read_csv(base + "f32_pmc_WRITE_SIZE/f32_counter_collection.csv")
    st_all = pd.read_csv(base + "f32_stats/f32_kernel_stats.csv")
    for key, alg in (("kres_sym_kernel", 4.0 * n * n), ("kbuild_mfma_kernel", 2.0 * n * (n + 1) + 4.0 * n * (n + 1) + 8.0 * n * 32)):
        sel2 = lambda df, col: df[df[col].str.contains(key, regex=False)]
        fk2, wk2, sk2 = sel2(fe_all, "Kernel_Name"), sel2(wr_all, "Kernel_Name"), sel2(st_all, "Name")
        if len(fk2) and len(sk2):
            ms = float(sk2["AverageNs"].iloc[0]) / 1e6
            fb, wb = float(fk2["Counter_Value"].mean()) * 1024 * 2, float(wk2["Counter_Value"].mean()) * 1024
            f32[key] = {"launches_pmc": int(len(fk2)), "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb, "algorithmic_bytes_per_launch": alg,
                        "rocprofv3_stats_avg_ms": ms, "hbm_total_GBps": (fb + wb) / (ms * 1e-3) / 1e9}
    json.dump(f32, open("profiles/%s_pmc_syrk128_f32.json" % rnd, "w"), indent=1)
    print("syrk128<float>: %.1f TFLOP/s in-kernel (%.2f of peak), fetch %.1f GB, write %.1f GB per run" % (f32["tflops_in_kernel"], f32["frac_of_fp32_mfma_peak"], f32["fetch_bytes_total_corrected"] / 1e9, f32["write_bytes_total"] / 1e9))
except FileNotFoundError as e:
    print("no fp32 passes:", e)
