"""Regenerate the measured tables of DESIGN.md (between the GENERATED markers) from profiles/<round>_*: every figure names the file it comes from.
    python tools/design_tables.py [r05]        (rewrites DESIGN.md in place; prints the tables)"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (R, name))
v = json.load(open(P("bench_line_verbose.json")))
pm = json.load(open(P("pmc_syrk128.json")))
kb = json.load(open(P("pmc_kbuild.json")))
f32 = json.load(open(P("pmc_syrk128_f32.json")))
stats = {r["Name"]: r for r in csv.DictReader(open(P("rocprofv3_kernel_stats.csv")))}
def stat(prefix):
    """calls and average ms over EVERY instantiation whose name contains `prefix` (the trailing update is two: with and without the diagonal / ride tile forms)"""
    calls, tot = 0, 0.0
    for k, r in stats.items():
        if prefix in k:
            calls += int(r["Calls"]); tot += int(r["Calls"]) * float(r["AverageNs"]) / 1e6
    return (calls, tot / calls) if calls else (0, float("nan"))
rf = v["roofline"]
oc = v["other_configs"]
def trace(name):
    """first line + per-kernel table of a trace summary"""
    try:
        return open(P("trace_%s.txt" % name)).read().splitlines()
    except OSError:
        return []
def trace_kernel(name, kern):
    for ln in trace(name):
        if ln.startswith(kern):
            f = ln.split()
            return int(f[-3]), float(f[-2]), float(f[-1])
    return None

out = []
A = out.append
A("<!-- BEGIN GENERATED: tables (tools/design_tables.py %s) -->" % R)
A("")
A("**Headline** (`profiles/%s_bench_line_verbose.json`, the default `python bench.py`, CPU protocol included; compact form `profiles/%s_bench_line.json`):" % (R, R))
A("")
A("| quantity | value | source |")
A("|---|---|---|")
G = int(v["config"]["fits_per_step"]); YR = int(v["config"].get("years_resident_per_rank", 40))
A("| GP fits/s, configs[2] (n = %d, d = %d; a step = %d fits in lockstep: the %d years at %g consecutive grid points) | **%.1f** (%.2f ms per %d-fit step; %.1f without the event brackets) | `value`, `ms_per_step`, `without_event_brackets` |"
  % (v["config"]["n"], v["config"]["d"], G, YR, G / YR, v["value"], v["ms_per_step"], G, v.get("without_event_brackets", {}).get("value", float("nan"))))
A("| whole fit, fraction of the fp64 MFMA peak (78.6 TFLOP/s) | %.3f (%.1f TFLOP/s) | `whole_fit_frac_of_fp64_mfma_peak` |" % (v["whole_fit_frac_of_fp64_mfma_peak"], v["whole_fit_tflops"]))
A("| `roofline`: `syrk128_kernel<double>` by HIP events over the timed region | **%.3f** = %.2f TFLOP/s; %d launches, avg %.3f ms, %.4g flop each | `roofline` |"
  % (rf["frac"], rf["achieved"], rf["launches"], rf["avg_launch_ms"], rf["flops_per_launch"]))
c, a = stat("syrk128_kernel<double, false, false")
A("| the same kernel in `rocprofv3 --kernel-trace --stats` of the same command | %d calls, avg %.3f ms (%.1f %% from the HIP-event average) | `profiles/%s_rocprofv3_kernel_stats.csv` |" % (c, a, 100 * abs(a - rf["avg_launch_ms"]) / rf["avg_launch_ms"], R))
if rf.get("strips_beside_update"):
    A("| ... with the strip solve beside the trailing update (`strips_after_update` = 0, the default up to r4) | %.3f (%.1f TFLOP/s), %.1f fits/s | `roofline.strips_beside_update` |"
      % (rf["strips_beside_update"]["frac"], rf["strips_beside_update"]["achieved"], rf["strips_beside_update"]["fits_per_s_with_this_schedule"]))
elif rf.get("unshared"):
    A("| ... when the strip solve does not share the chip with it (`roofline.unshared`) | %.3f (%.1f TFLOP/s) | `roofline.unshared` |" % (rf["unshared"]["frac"], rf["unshared"]["achieved"]))
A("| HBM-side traffic of that kernel per launch (PMC, separate passes) | %.2f GB fetched (FETCH_SIZE x 2) + %.2f GB written = **%.2f GB**; algorithmic C bytes %.2f GB; matrix pipe busy %.3f | `profiles/%s_pmc_syrk128.json` |"
  % (pm["fetch_bytes_per_launch_corrected"] / 1e9, pm["write_bytes_per_launch"] / 1e9, pm["traffic_bytes_per_launch"] / 1e9, rf.get("algorithmic_c_bytes_per_launch", 0) / 1e9, pm["mfma_pipe_busy_fraction"], R))
cb = v["cpu_baseline"]
A("| CPU baseline (oracle in the reference's call sequence, %d BLAS threads, 1 warm-up + 3 timed, median) | %.4f fits/s (%.1f s per fit); best practice %.1f s; GPU / CPU = %.0f (context, not credit) | `cpu_baseline`, `cpu_baseline_best_practice` |"
  % (cb["cores"], cb["value"], cb["seconds"], v["cpu_baseline_best_practice"]["seconds"], v["vs_cpu_baseline"]))
pr = v["parity"]
A("| parity of the first timed step vs the oracle (mean / variance / nlML, relative) | %.1e / %.1e / %.1e (tolerance 1e-8) | `parity` |" % (pr["batch_step0_mean_rel"], pr["batch_step0_var_rel"], pr["batch_step0_nlml_rel"]))
A("| covariance build `kbuild_kernel<double,8>`, %d members per launch | %.3f ms; %.2f GB written + %.2f GB fetched (PMC) = %.2f TB/s = %.2f of 8 TB/s; algorithmic %.2f GB -> %.2f TB/s | `profiles/%s_pmc_kbuild.json` |"
  % (G, kb["rocprofv3_stats_avg_ms"], kb["write_bytes_per_launch"] / 1e9, kb["fetch_bytes_per_launch_corrected"] / 1e9, kb["hbm_total_GBps"] / 1e3, kb["frac_of_8TBps"], kb["algorithmic_bytes_per_launch"] / 1e9,
     kb["algorithmic_bytes_per_launch"] / (kb["rocprofv3_stats_avg_ms"] * 1e-3) / 1e12, R))
A("| `syrk128_kernel<float>` over all launches of a configs[4] fit | %.1f TFLOP/s = %.3f of 157.3; matrix pipe busy %s | `profiles/%s_pmc_syrk128_f32.json` |"
  % (f32["tflops_in_kernel"], f32["frac_of_fp32_mfma_peak"], ("%.3f" % f32["mfma_pipe_busy_fraction"]) if "mfma_pipe_busy_fraction" in f32 else "n/a", R))
A("")
A("**Other BASELINE configurations on one GPU** (untimed extras of the same run, `other_configs` of `profiles/%s_bench_line_verbose.json`; kernel-trace summaries `profiles/%s_trace_*.txt`):" % (R, R))
A("")
A("| configuration | measured | where the time goes (trace) |")
A("|---|---|---|")
def oc_get(prefix):
    for k, x in oc.items():
        if k.startswith(prefix):
            return x
    return {}
c1, c3, c4, c4g = oc_get("configs[1]"), oc_get("configs[3]"), oc_get("configs[4] n="), oc_get("configs[4] shape")
def tk(name, kern, label):
    t = trace_kernel(name, kern)
    return "%s %d x %.1f us" % (label, t[0], t[2]) if t else ""
A("| configs[1] n = 4096, d = 8 fp64 RBF, single fit | **%.3f ms** (%.1f fits/s, %.3f of peak) | %s; %s; %s; %s (`%s_trace_single_fit_n4096.txt`, outer width 8) |"
  % (c1.get("ms_per_fit", 0), c1.get("fits_per_s", 0), c1.get("frac_of_peak", 0), tk("single_fit_n4096", "diag_update_kernel<double>", "diagonal block + ride"), tk("single_fit_n4096", "chain_link_kernel<double>", "link"),
     tk("single_fit_n4096", "potrf_diag_kernel<double>", "first block of a panel"), tk("single_fit_n4096", "gemm_mfma_kernel<double, 64, 64", "panel-boundary updates"), R))
A("| n = 8192, d = 8, single fit | %s | %s; %s; %s |" % ((trace("single_fit_n8192") or ["?"])[0], tk("single_fit_n8192", "diag_update_kernel<double>", "diagonal block + ride"), tk("single_fit_n8192", "chain_link_kernel<double>", "link"),
                                                  tk("single_fit_n8192", "syrk128_kernel<double, false", "trailing updates")))
A("| configs[3] n = 16384, d = 16 fp64 RBF, single fit on one GPU | **%.2f ms** (%.3f of peak) | %s; %s; %s; %s |"
  % (c3.get("ms_per_fit", 0), c3.get("frac_of_peak", 0), tk("single_fit_n16384", "syrk128_kernel<double, false", "trailing updates"), tk("single_fit_n16384", "diag_update_kernel<double>", "diagonal block + ride"),
     tk("single_fit_n16384", "chain_link_kernel<double>", "link"), tk("single_fit_n16384", "potrf_diag_kernel<double>", "plain diagonal blocks")))
A("| configs[4] n = 32768, d = 32 fp32 Matern-5/2 + fp64 refinement, single fit | **%.1f ms** (%.3f of the fp32 peak; refinement residual %.1e); lockstep group of 4: %.1f ms per fit (%.3f) | %s; %s; %s |"
  % (c4.get("ms_per_fit", 0), c4.get("frac_of_peak", 0), c4.get("refinement_residual", 0), c4g.get("ms_per_fit", 0), c4g.get("frac_of_peak", 0), tk("fp32_n32768", "syrk128_kernel<float, false", "trailing updates"),
     tk("fp32_n32768", "kres_lower_rows_kernel", "residual row pass"), tk("fp32_n32768", "kbuild_mfma_kernel<float, 32>", "covariance build")))
ml = oc.get("mlii", {})
A("| MLII (nlML + exact gradient) | single fit: %.2f ms at n = 4096, %.2f ms at n = 8192 (%.2f of peak); lockstep group of 40: %.2f ms per evaluation at n = 4096 (%.2f), **%.2f ms at n = 8192 (%.2f)** | |"
  % (ml.get("n=4096", {}).get("ms", 0), ml.get("n=8192", {}).get("ms", 0), ml.get("n=8192", {}).get("frac_of_fp64_mfma_peak", 0), ml.get("n=4096 lockstep group of 40", {}).get("ms_per_evaluation", 0),
     ml.get("n=4096 lockstep group of 40", {}).get("frac_of_fp64_mfma_peak", 0), ml.get("n=8192 lockstep group of 40", {}).get("ms_per_evaluation", 0), ml.get("n=8192 lockstep group of 40", {}).get("frac_of_fp64_mfma_peak", 0)))
rk = oc.get("reference_kernel_grid", {})
A("| reference kernel at reference size: 3 regions x 40 years x 20 x 20 grid = %d fits, one launch | %.2f ms per call (%.1f M fits/s incl. copies) vs %.1f k fits/s for the oracle's loop on the host | |"
  % (rk.get("fits", 0), rk.get("ms_per_launch", 0), rk.get("fits_per_s", 0) / 1e6, rk.get("cpu_oracle_loop", {}).get("fits_per_s", 0) / 1e3))
rm = oc.get("reference_kernel_mlii", {})
if rm and "evaluations_per_s" in rm:
    op = rm.get("optimise_120_region_years", {})
    A("| the reference's MLII closure for its own kernel, batched: %d evaluations (value + reference \"gradient\" + exact gradient), one launch | %.2f ms per call (**%.1f M evaluations/s**); the optimiser of `:259-262` for 120 (region, year) data sets in lockstep: %d launches, %.0f ms, %d converged | |"
      % (rm.get("evaluations", 0), rm.get("ms_per_launch", 0), rm.get("evaluations_per_s", 0) / 1e6, op.get("launches", 0), 1e3 * op.get("seconds", 0), op.get("converged", 0)))
A("")
try:
    ch = json.load(open(P("pmc_chain_kernels.json")))
    A("**The panel stream's kernels** in the same bench command (`profiles/%s_pmc_chain_kernels.json`: PMC passes serialise kernels, so the counters are each kernel ALONE on the chip; the durations are from `--kernel-trace --stats` of the normal run, beside the trailing update):" % R)
    A("")
    A("| kernel | launches per command | avg ms (beside the update) | total ms | MFMA pipe busy | CU busy | LDS bank conflicts / LDS-active cycles | HBM bytes per launch (fetch x 2 + write) |")
    A("|---|---|---|---|---|---|---|---|")
    names = {"diag_update_kernel<double": "`diag_update_kernel` (diagonal blocks + riding in-panel updates)", "panel_strip_kernel<double": "`panel_strip_kernel` (rows below a panel's top block)",
             "chain_link_kernel<double": "`chain_link_kernel`", "potrf_diag_kernel<double": "`potrf_diag_kernel` (first block of a panel)", "syrk128_kernel<double, true": "`syrk128_kernel<double, SET>` (column solves)"}
    for k, e in ch["kernels"].items():
        A("| %s | %d | %.3f | %.1f | %s | %s | %s | %.2f GB |" % (names.get(k, k), e["rocprofv3_stats_calls"], e["rocprofv3_stats_avg_ms"], e["rocprofv3_stats_total_ms"],
          ("%.2f" % e["mfma_pipe_busy_fraction"]) if e.get("mfma_pipe_busy_fraction") is not None else "n/a", ("%.2f" % e["cu_busy_fraction"]) if "cu_busy_fraction" in e else "n/a",
          ("%.2f" % e["lds_bank_conflict_fraction"]) if "lds_bank_conflict_fraction" in e else "n/a", (e["fetch_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]) / 1e9))
    A("")
except OSError:
    pass
try:
    cp = json.load(open(P("critical_path.json")))
    ui = cp["update_queue_idle_ms"]
    A("**Where the update queue idles inside one step** (`profiles/%s_critical_path.json`, from the kernel trace of `profiles/%s_trace_batch_group.txt`): step %.1f ms, update queue busy %.1f ms, idle %.1f ms = head %.1f (covariance build on the same queue excluded; first panel: %s) + between trailing updates %.1f (late panels: %s) + tail %.1f."
      % (R, R, cp["step_ms"], cp["update_queue_busy_ms"], ui["total"], ui["head"], ", ".join("%s %.1f" % kv for kv in list(cp["other_queue_during_idle_ms"]["head"].items())[:3]), ui["between"],
         ", ".join("%s %.1f" % kv for kv in list(cp["other_queue_during_idle_ms"]["between"].items())[:3]), ui["tail"]))
    A("")
except OSError:
    pass
A("<!-- END GENERATED -->")
txt = "\n".join(out)
print(txt)
dp = os.path.join(ROOT, "DESIGN.md")
d = open(dp).read()
if "<!-- BEGIN GENERATED" in d:
    d = re.sub(r"<!-- BEGIN GENERATED.*?<!-- END GENERATED -->", lambda m: txt, d, flags=re.S)
    open(dp, "w").write(d)
