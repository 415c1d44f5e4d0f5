#!/usr/bin/env python3
"""Per-tick kernel breakdown from a rocprofv3 --kernel-trace CSV."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ticks, cur = [], None
for r in rows:
    m = re.search(r'\b(k_[a-z_]+)', r['Kernel_Name'])
    if not m: continue
    n = m.group(1)
    if n == 'k_census_finish': n = 'k_finish'      # round 5: census and epilogue in one launch
    if 'init' in n or 'inject' in n or 'commit' in n or n == 'k_deliver_list': continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if n == 'k_pending' or (n == 'k_begin' and (cur is None or 'k_deliver' in cur)): cur = {'start': int(r['Start_Timestamp'])}
    if cur is None: continue
    cur[n] = cur.get(n, 0) + d
    if n == 'k_finish': cur['span'] = (int(r['End_Timestamp']) - cur['start']) / 1e3; ticks.append(cur); cur = None
ticks = ticks[skip:]
print(len(ticks), 'ticks')
for k in ['k_pending', 'k_begin', 'k_deliver', 'k_resolve', 'k_census', 'k_finish', 'span']:
    v = sorted(t.get(k, 0) for t in ticks)
    print(f"{k:10s} min {v[0]:7.1f} med {v[len(v)//2]:7.1f} p75 {v[int(len(v)*.75)]:7.1f} p90 {v[int(len(v)*.9)]:7.1f} max {v[-1]:7.1f} sum {sum(v)/1e3:7.2f} ms")
for t in sorted(ticks, key=lambda t: -t['span'])[:5]: print({k: round(v, 1) for k, v in t.items() if k != 'start'})
