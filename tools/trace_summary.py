"""Timeline summary of one fit from a rocprofv3 kernel trace: per-queue busy time, kernels by class, and a coarse timeline
(what runs in each 1/40 of the fit).  Usage: trace_summary.py <kernel_trace.csv> [first-kernel-name-prefix]"""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp')
df['name'] = df['Kernel_Name'].str.replace('void ', '').str.replace('sigp::', '').str.slice(0, 40)
mark = sys.argv[2] if len(sys.argv) > 2 else 'kbuild'
kb = df[df['name'].str.startswith(mark)]
t0, t1 = kb['Start_Timestamp'].iloc[-2], kb['Start_Timestamp'].iloc[-1]
fit = df[(df['Start_Timestamp'] >= t0) & (df['Start_Timestamp'] < t1)].copy()
fit['dur'] = fit['End_Timestamp'] - fit['Start_Timestamp']
T = (t1 - t0)
print("fit %.2f ms, %d kernels" % (T / 1e6, len(fit)))
g = fit.groupby('name')['dur'].agg(['count', 'sum', 'mean'])
print(g.assign(sum=g['sum'] / 1e6, mean=g['mean'] / 1e3).sort_values('sum', ascending=False).head(12).to_string())
for q, fq in fit.groupby('Queue_Id'):
    print("queue %s: busy %.2f ms (%d kernels)" % (q, fq['dur'].sum() / 1e6, len(fq)))
nb = 40
print("timeline (%d bins): per bin, busy share of each queue" % nb)
for b in range(nb):
    a, e = t0 + T * b / nb, t0 + T * (b + 1) / nb
    row = []
    for q, fq in fit.groupby('Queue_Id'):
        ov = (fq['End_Timestamp'].clip(upper=e) - fq['Start_Timestamp'].clip(lower=a)).clip(lower=0).sum()
        row.append("q%s %3.0f%%" % (q, 100 * ov / (e - a)))
    print("  %5.1f ms  %s" % ((a - t0) / 1e6, "  ".join(row)))
