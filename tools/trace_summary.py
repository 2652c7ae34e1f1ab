"""Timeline summary of one fit from a rocprofv3 kernel trace: per-queue busy time, kernels by class, and a coarse timeline
(what runs in each 1/40 of the fit).  Usage: trace_summary.py <kernel_trace.csv> [first-kernel-name-prefix] [--critical-path out.json]
--critical-path: where the queue of the dominant kernel (the update stream) is IDLE inside the step, and what the other queue runs meanwhile:
head (before the first trailing update), between updates (the panel stream is the critical path), tail (after the last one)."""
import json
import sys
import pandas as pd
cp_out = None
if "--critical-path" in sys.argv:
    i = sys.argv.index("--critical-path")
    cp_out = sys.argv[i + 1]
    del sys.argv[i:i + 2]
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

if cp_out:
    # the update queue is the one the step's covariance build runs on (in-panel updates on the panel stream use the same tile kernel)
    qd = fit[fit['name'].str.startswith(mark)]['Queue_Id'].iloc[0]
    upd = fit[fit['Queue_Id'] == qd].sort_values('Start_Timestamp')
    dom = upd[upd['name'].str.startswith('syrk128_kernel<double, false') | upd['name'].str.startswith('syrk128_kernel<float, false')]
    oth = fit[fit['Queue_Id'] != qd]
    gaps = []          # (start, end, kind)
    first, last = dom['Start_Timestamp'].min(), dom['End_Timestamp'].max()
    cur = t0
    for _, k in upd.iterrows():
        if k['Start_Timestamp'] > cur:
            a, e = cur, k['Start_Timestamp']
            gaps.append((a, e, 'head' if e <= first else 'tail' if a >= last else 'between'))
        cur = max(cur, k['End_Timestamp'])
    if t1 > cur:
        gaps.append((cur, t1, 'tail'))
    rec = {"step_ms": T / 1e6, "update_queue_busy_ms": float(upd['dur'].sum() / 1e6), "update_queue_idle_ms": {}, "other_queue_during_idle_ms": {}}
    for kind in ('head', 'between', 'tail'):
        gk = [(a, e) for a, e, kk in gaps if kk == kind]
        rec["update_queue_idle_ms"][kind] = float(sum(e - a for a, e in gk) / 1e6)
        by = {}
        for a, e in gk:
            ov = (oth['End_Timestamp'].clip(upper=e) - oth['Start_Timestamp'].clip(lower=a)).clip(lower=0)
            for nm, v in ov.groupby(oth['name']).sum().items():
                if v > 0:
                    by[nm.split('(')[0]] = by.get(nm.split('(')[0], 0.0) + v / 1e6
        rec["other_queue_during_idle_ms"][kind] = {k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:6]}
    rec["update_queue_idle_ms"]["total"] = float(sum(rec["update_queue_idle_ms"].values()))
    rec["note"] = ("one lockstep step of the bench from a rocprofv3 kernel trace: the update queue (syrk128 trailing updates, covariance build, epilogue) is idle while the panel "
                   "stream (diagonal blocks with riding in-panel updates, strip solves) is the critical path: head = build + first panel, between = late panels whose strips and "
                   "in-panel updates outlast the shrinking trailing update, tail = the last panel")
    json.dump(rec, open(cp_out, "w"), indent=1)
    print("critical path:", json.dumps(rec["update_queue_idle_ms"]))
