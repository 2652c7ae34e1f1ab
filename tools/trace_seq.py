"""Kernel sequence of one fit from a rocprofv3 kernel trace (start offset, duration, queue, grid), between the last two covariance builds:
trace_seq.py <kernel_trace.csv> [first_us last_us]"""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp')
df['name'] = df['Kernel_Name'].str.replace('void ', '').str.replace('sigp::', '').str.slice(0, 44)
kb = df[df['name'].str.startswith('kbuild')]
t0, t1 = kb['Start_Timestamp'].iloc[-2], kb['Start_Timestamp'].iloc[-1]
fit = df[(df['Start_Timestamp'] >= t0) & (df['Start_Timestamp'] < t1)]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
prev_end = {}
for _, r in fit.iterrows():
    st = (r['Start_Timestamp'] - t0) / 1e3
    if st < lo or st > hi:
        continue
    q = r['Queue_Id']
    gap = (r['Start_Timestamp'] - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = r['End_Timestamp']
    print("%9.1f us  q%s  gap %6.1f  dur %7.1f us  grid %6d x %3d  %s" % (st, q, gap, (r['End_Timestamp'] - r['Start_Timestamp']) / 1e3, r['Grid_Size_X'] // max(1, r['Workgroup_Size_X']), r['Grid_Size_Y'], r['name']))
