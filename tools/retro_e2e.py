"""A whole retrospective run (north/retrospective_forecasts/September1st_retro.py:284-299: detrend -> networks -> forecast -> skill)
on synthetic inputs of the real shape -- a 57 x 57 ice-concentration grid, 1979..fmax, three regions -- through this package, stage
by stage: where the wall time goes once the GP itself takes milliseconds.
    python tools/retro_e2e.py [fmin] [fmax] [--host]      (--host: tau / area sums / detrending on the host instead of the GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.stats import linregress
import seaiceextentforecasting_amd as S
import seaiceextentforecasting_amd.networks as NW

args = [a for a in sys.argv[1:] if not a.startswith("--")]
fmin = int(args[0]) if len(args) > 0 else 1985
fmax = int(args[1]) if len(args) > 1 else 2019
on_host = "--host" in sys.argv
T = fmax - 1979 + 1
rng = np.random.default_rng(2024)
# synthetic June-mean concentration anomalies: six coherent regions + noise inside a disc, land / open ocean outside (NaN)
base = np.cumsum(rng.standard_normal((6, T)), axis=1) * 0.3 + rng.standard_normal((6, T))
sic = np.full((57, 57, T), np.nan)
for i in range(57):
    for j in range(57):
        if (i - 28) ** 2 + (j - 28) ** 2 < 26 ** 2:
            sic[i, j] = 0.5 + 0.1 * base[(i // 20) * 2 + (j // 30)] + 0.07 * rng.standard_normal(T) - 0.002 * np.arange(T)
SIC = {"data": sic, "psar": rng.uniform(0.8, 1.2, (57, 57))}
regions = ["Pan-Arctic", "Beaufort", "Chukchi"]
SIEs, SIEs_dt, SIEs_trend = {}, {}, {}
for r, scale in zip(regions, (6.0, 0.6, 0.5)):
    SIEs[r] = (scale - 0.01 * scale * np.arange(T) + 0.05 * scale * (base[regions.index(r)] + rng.standard_normal(T))).round(3)
    trend = np.zeros((fmax - (fmin - 1) + 1, 2)); dt = np.zeros((fmax - (fmin - 1) + 1, T))      # read_SIE, :60-69
    for year in range(fmin - 1, fmax + 1):
        n = year - 1979 + 1
        reg = linregress(np.arange(n), SIEs[r][:n])
        trend[year - (fmin - 1)] = reg[0], reg[1]
        dt[year - (fmin - 1), :n] = SIEs[r][:n] - (reg[0] * np.arange(n) + reg[1])
    SIEs_trend[r], SIEs_dt[r] = trend, dt.round(3)

with S.GPR(kernel="netdiffusion") as gp:
    eng = None if on_host else gp
    t0 = time.perf_counter()
    S.detrend(SIC, fmin, fmax, engine=eng)
    t1 = time.perf_counter()
    NW.networks_retro(SIC, fmin, fmax, engine=eng)
    t2 = time.perf_counter()
    out = S.retro_forecast("north_September", SIC, SIEs_dt, SIEs_trend, fmin, fmax, gp=gp, batched=True)
    t3 = time.perf_counter()
    sk_rt, sk_dt, _ = S.skill(out, SIEs, SIEs_dt, fmin, fmax, regions)
    t4 = time.perf_counter()
ny = fmax - fmin + 1
print("retro run %d-%d (%d years x %d regions = %d GP fits, %d networks on a 57x57 grid, %s):" % (fmin, fmax, ny, len(regions), ny * len(regions), ny,
      "tau / sums / detrend on the host" if on_host else "tau / sums / detrend on the GPU"))
print("  detrend (every cut-off year)   %7.3f s" % (t1 - t0))
print("  networks (tau, areas, series)  %7.3f s   (%.3f s per year)" % (t2 - t1, (t2 - t1) / ny))
print("  forecast (features + GP batch) %7.3f s" % (t3 - t2))
print("  skill                          %7.3f s" % (t4 - t3))
print("  total                          %7.3f s;  skill (with trend) %s" % (t4 - t0, [float(x) for x in sk_rt]))
