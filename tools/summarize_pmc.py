"""Summarise the rocprofv3 --pmc passes of tools/collect_profiles.sh into profiles/<round>_pmc_syrk128.json."""
import json, sys
import pandas as pd
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
base = "gpurun_out/%s/" % rnd
out = {"command": "rocprofv3 --pmc <C> --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --no-profile  (one pass per counter set)",
       "kernel": "sigp::syrk128_kernel<double, false, false>", "notes": []}
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
kb = {"kernel": "sigp::kbuild_kernel<double, 8> (RBF covariance build of 40 lockstep members, lower 64x128 tiles, n = 8192, d = 8)",
      "launches_pmc": int(len(wk)), "write_bytes_per_launch": float(wk["Counter_Value"].mean()) * 1024,
      "fetch_bytes_per_launch_corrected": float(fk["Counter_Value"].mean()) * 1024 * 2,
      "algorithmic_bytes_per_launch": 40 * (4.0 * 8192 * 8193 + 8.0 * 8192 * 8),
      "rocprofv3_stats_avg_ms": float(st["AverageNs"].iloc[0]) / 1e6, "rocprofv3_stats_calls": int(st["Calls"].iloc[0])}
kb["hbm_write_GBps"] = kb["write_bytes_per_launch"] / (kb["rocprofv3_stats_avg_ms"] * 1e-3) / 1e9
kb["hbm_total_GBps"] = (kb["write_bytes_per_launch"] + kb["fetch_bytes_per_launch_corrected"]) / (kb["rocprofv3_stats_avg_ms"] * 1e-3) / 1e9
kb["frac_of_8TBps"] = kb["hbm_total_GBps"] / 8000.0
json.dump(kb, open("profiles/%s_pmc_kbuild.json" % rnd, "w"), indent=1)
print("kbuild: write GB %.2f fetch GB %.2f in %.3f ms -> %.2f TB/s" % (kb["write_bytes_per_launch"] / 1e9, kb["fetch_bytes_per_launch_corrected"] / 1e9, kb["rocprofv3_stats_avg_ms"], kb["hbm_total_GBps"] / 1e3))
