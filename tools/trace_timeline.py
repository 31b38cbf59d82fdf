"""Timeline of the last render in a rocprofv3 kernel trace: how long 0 / 1 / 2 / ... kernels ran side by side, per-queue gaps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
res = [i for i, r in enumerate(rows) if 'k_resolve' in r['Kernel_Name']]
seg = rows[res[-2] + 1:res[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
ev = []
for r in seg:
    n = r['Kernel_Name']
    n = 'ext' if 'k_extend' in n else 'shade' if 'k_shade' in n else 'tail' if 'k_tail' in n else 'res' if 'resolve' in n else 'misc'
    ev.append((n, r['Queue_Id'], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3))
pts = sorted([(s, 1) for _, _, s, e in ev] + [(e, -1) for _, _, s, e in ev])
cur, last, by = 0, 0, {}
for t, d in pts:
    by[cur] = by.get(cur, 0) + t - last
    last = t
    cur += d
print("total %.0f us; time with n kernels running:" % ev[-1][3], {k: round(v) for k, v in sorted(by.items())})
for q in sorted(set(e[1] for e in ev)):
    qs = [e for e in ev if e[1] == q and e[0] in ('ext', 'shade', 'tail')]
    if not qs: continue
    gaps = [qs[i + 1][2] - qs[i][3] for i in range(len(qs) - 1)]
    print("queue", q, "launches", len(qs), "gaps %.0f us" % sum(gaps), "first start %.0f last end %.0f" % (qs[0][2], qs[-1][3]))
    print("   ", " ".join("%s%.0f" % (e[0][0], e[3] - e[2]) for e in qs))
