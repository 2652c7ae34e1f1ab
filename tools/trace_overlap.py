"""Summarise a rocprofv3 kernel trace: when does each fit's kbuild start, and how many chains overlap."""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp')
df['name'] = df['Kernel_Name'].str.replace('void ', '').str.replace('sigp::', '').str.slice(0, 40)
kb = df[df['name'].str.startswith('kbuild')]
t0 = kb['Start_Timestamp'].iloc[0]
print("kbuild starts (ms, queue):", [(round((t - t0) / 1e6, 2), q) for t, q in zip(kb['Start_Timestamp'], kb['Queue_Id'])])
d = df[df['name'].str.startswith('potrf_diag')]
ev = sorted([(s, 1) for s in d['Start_Timestamp']] + [(e, -1) for e in d['End_Timestamp']])
cur = 0; last = ev[0][0]; hist = {}
for t, x in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += x
tot = sum(hist.values())
print("diag-kernel concurrency:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
