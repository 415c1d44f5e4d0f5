#!/usr/bin/env python3
"""What the bounded memberlist queue costs BASELINE config #4's answer ("a property of the 32-entry queue, not of memberlist").
memberlist's TransmitLimitedQueue is unbounded; the product library holds queue_cap <= 32 entries per node (LDS) with Prune() semantics.  The
CHECKER can hold up to 4 096, so the same population — N nodes, 5 % stopped at once, run until every survivor holds every victim dead — is run on it
with caps from 8 to 4 096 (at these N a queue never reaches 4 096: as good as unbounded) and the seconds to full detection compared.
  python tools/queue_cap_sweep.py --nodes 8192 [--caps 8,16,32,64,128,256,4096]   (one process per cap, in parallel)"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=8192)
ap.add_argument("--caps", default="8,16,32,64,128,256,4096")
ap.add_argument("--seconds", type=int, default=900)
a = ap.parse_args()
procs = {}
for cap in [int(c) for c in a.caps.split(",")]:
    procs[cap] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "config4_run.py"), "--oracle", "--nodes", str(a.nodes), "--queue-cap", str(cap),
                                   "--seconds", str(a.seconds), "--every", "5"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(f"config #4's shape on the checker: {a.nodes} nodes, {int(a.nodes * 0.05)} stopped at t = 1 s, LAN timers, k = 3; full detection = every survivor holds every victim dead")
print(f"{'queue_cap':>9} {'full detection at (s)':>22} {'queue_drops':>14} {'msgs applied':>14} {'drops / applied':>16}")
for cap, p in procs.items():
    out = p.communicate()[0].splitlines()
    rows = [json.loads(l) for l in out if l.startswith("{")]
    done = next((r for r in rows if r["pairs"] and r["dead"] == r["pairs"]), None)
    last = done or (rows[-1] if rows else None)
    if not last:
        print(f"{cap:9d} failed: {out[-1] if out else ''}"); continue
    ap_ = sum(last["applied"])
    print(f"{cap:9d} {(str(last['t_s']) if done else '> ' + str(last['t_s'])):>22} {last['queue_drops']:14d} {ap_:14d} {last['queue_drops'] / max(ap_, 1):16.3f}", flush=True)
