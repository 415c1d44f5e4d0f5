#!/usr/bin/env python3
"""Per-tick kernel durations (us) along the run, from a rocprofv3 --kernel-trace CSV: one line per tick."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
tick, cur = 0, {}
for r in rows:
    m = re.search(r'\b(k_[a-z_]+)', r['Kernel_Name'])
    if not m: continue
    n = m.group(1)
    if n == 'k_census_finish': n = 'k_finish'      # round 5: census and epilogue in one launch
    if n not in ('k_begin', 'k_deliver', 'k_resolve', 'k_census', 'k_finish', 'k_pending'): continue
    cur[n] = cur.get(n, 0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if n == 'k_finish':
        print(f"tick {tick:4d} " + " ".join(f"{k[2:]} {cur.get(k, 0):6.1f}" for k in ('k_begin', 'k_deliver', 'k_resolve', 'k_census', 'k_finish')))
        tick += 1; cur = {}
