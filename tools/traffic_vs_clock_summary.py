"""Table for tools/traffic_vs_clock.sh: per tile walk -- fits/s, per-launch fraction of peak, mean sclk / package power while the bench ran, L2-miss fetch per launch."""
import glob, json, os, re, sys
import pandas as pd
out = sys.argv[1]
print("walk (xcd_chunks P) | fits/s (2 runs) | roofline.frac | sclk MHz mean (samples) | power W mean | FETCH GB per syrk128 launch (x2 corrected)")
for P in (0, 4, 8, 16):
    vals, fr, clk, pw = [], [], [], []
    for rep in (1, 2):
        try:
            d = json.loads([l for l in open("%s/bench_P%d_%d.json" % (out, P, rep)) if l.startswith("{")][-1])
            vals.append(d["value"]); fr.append(d["roofline"]["frac"])
        except Exception as e:
            vals.append(float("nan")); fr.append(float("nan"))
        try:
            txt = open("%s/clock_P%d_%d.txt" % (out, P, rep)).read()
            c = [float(x) for x in re.findall(r"sclk.*?\((\d+)Mhz\)", txt)]
            w = [float(x) for x in re.findall(r"Power \(W\):\s*([\d.]+)", txt)]
            busy = [(a, b) for a, b in zip(c, w) if b > 600]           # samples taken while the chip was under load
            clk += [a for a, _ in busy]; pw += [b for _, b in busy]
        except Exception:
            pass
    fetch = float("nan")
    try:
        f = glob.glob("%s/pmc_P%d/**/*counter_collection.csv" % (out, P), recursive=True)[0]
        df = pd.read_csv(f)
        df = df[df["Kernel_Name"].str.contains("syrk128_kernel<double, false", regex=False)]
        fetch = df["Counter_Value"].mean() * 1024 * 2 / 1e9
    except Exception as e:
        pass
    m = lambda a: sum(a) / len(a) if a else float("nan")
    print("P=%2d | %s | %s | %.0f (%d) | %.0f | %.2f" % (P, " / ".join("%.1f" % v for v in vals), " / ".join("%.3f" % v for v in fr), m(clk), len(clk), m(pw), fetch))
