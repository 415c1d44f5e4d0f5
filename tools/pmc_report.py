#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_pass.sh): per kernel, counters of the heaviest dispatches."""
import collections, csv, glob, sys
root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
data = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> dispatch idx -> counter -> value
dur = collections.defaultdict(dict)
for p in sorted(glob.glob(f"{root}/pass*/*/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(p)))
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(p.replace("counter_collection", "kernel_trace")))}
    seq = collections.Counter()
    seen = {}
    for r in rows:
        import re as _re
        _m = _re.search(r"\b(k_[a-z_]+)", r["Kernel_Name"]); k = _m.group(1) if _m else r["Kernel_Name"]
        did = r["Dispatch_Id"]
        if did not in seen:
            seen[did] = seq[k]; seq[k] += 1
        idx = seen[did]
        data[k][idx][r["Counter_Name"]] = data[k][idx].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        t = trace.get(did)
        if t: dur[k][idx] = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
for k in ("k_begin", "k_deliver", "k_resolve", "k_census", "k_gossip_iq", "k_piggy_iq"):
    if k not in data: continue
    idxs = sorted(data[k], key=lambda i: -dur[k].get(i, 0))[:top]
    print(f"== {k}: mean over the {len(idxs)} longest dispatches (profiled durations)")
    agg = collections.defaultdict(float)
    for i in idxs:
        for c, v in data[k][i].items(): agg[c] += v / len(idxs)
    d = sum(dur[k].get(i, 0) for i in idxs) / len(idxs)
    print(f"   duration_us {d:9.1f}")
    for c in sorted(agg): print(f"   {c:22s} {agg[c]:16.1f}")
    if "FETCH_SIZE" in agg: print(f"   fetch_MB(x2 corr) {agg['FETCH_SIZE']*1024*2/1e6:9.1f}   raw {agg['FETCH_SIZE']*1024/1e6:9.1f}")
    if "WRITE_SIZE" in agg: print(f"   write_MB          {agg['WRITE_SIZE']*1024/1e6:9.1f}")
