#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_pass.sh.

usage: tools/pmc_traffic.py <pmc dir> <virtual nodes of the workload> [first N launches] > profiles/pmc_traffic.json
HBM bytes per launch = mean over every launch of the kernel in the bench run; FETCH_SIZE is in KiB-ish units of
64 B requests and is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests counted as 64 B).
"""
import collections, csv, glob, json, re, sys
root, nodes = sys.argv[1], int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # only the first N launches of each kernel (the bench's main run)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for p in sorted(glob.glob(f"{root}/pass*/*/*_counter_collection.csv")):
    rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Dispatch_Id"]))
    order = collections.defaultdict(dict)
    for r in rows:
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"): continue
        m = re.search(r"\b(k_[a-z_]+)", r["Kernel_Name"])
        if not m or m.group(1) not in ("k_begin", "k_deliver", "k_resolve", "k_census", "k_finish", "k_census_finish", "k_gossip_iq", "k_piggy_iq", "k_inbox_claim", "k_inbox_file"): continue
        k = m.group(1)
        idx = order[k].setdefault(r["Dispatch_Id"], len(order[k]))
        if first and idx >= first: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {}
for k, c in acc.items():
    nf, nw = len(launches[(k, "FETCH_SIZE")]) or 1, len(launches[(k, "WRITE_SIZE")]) or 1
    fetch, write = c["FETCH_SIZE"] * 1024 / nf, c["WRITE_SIZE"] * 1024 / nw
    out[k] = {"launches": nf, "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": 2 * fetch + write,
              "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncalibrated",
              "workload_nodes": nodes}
json.dump(out, sys.stdout, indent=1)
