"""Where a latency-bound single fit spends its time: per-kernel-class totals, launch gaps on the critical (panel) stream and the
kernel sequence between two consecutive diagonal-block kernels, from a rocprofv3 kernel trace of tools/single_trace.py."""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp')
df['name'] = df['Kernel_Name'].str.replace('void ', '').str.replace('sigp::', '').str.slice(0, 34)
kb = df[df['name'].str.startswith('kbuild')]
t0, t1 = kb['Start_Timestamp'].iloc[-2], kb['Start_Timestamp'].iloc[-1]      # one whole fit between two builds
fit = df[(df['Start_Timestamp'] >= t0) & (df['Start_Timestamp'] < t1)]
print("fit: %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(fit)))
g = fit.assign(dur=fit['End_Timestamp'] - fit['Start_Timestamp']).groupby('name')['dur'].agg(['count', 'sum', 'mean'])
print((g.assign(sum=g['sum'] / 1e3, mean=g['mean'] / 1e3).sort_values('sum', ascending=False)).head(10).to_string())
d = fit[fit['name'].str.startswith('potrf_diag')]
starts = list(d['Start_Timestamp']); ends = list(d['End_Timestamp'])
gaps = [(starts[i + 1] - ends[i]) / 1e3 for i in range(len(starts) - 1)]
print("diag kernels: %d, avg duration %.1f us, avg time between end of one and start of the next %.1f us (min %.1f, max %.1f)" % (
    len(d), (d['End_Timestamp'] - d['Start_Timestamp']).mean() / 1e3, sum(gaps) / len(gaps), min(gaps), max(gaps)))
i = len(starts) // 2
link = fit[(fit['Start_Timestamp'] >= ends[i]) & (fit['Start_Timestamp'] <= starts[i + 2])]
for _, r in link.iterrows():
    print("  +%7.1f us  %-36s %6.1f us  queue %s grid %s" % ((r['Start_Timestamp'] - ends[i]) / 1e3, r['name'], (r['End_Timestamp'] - r['Start_Timestamp']) / 1e3, r['Queue_Id'], r['Grid_Size_X']))
