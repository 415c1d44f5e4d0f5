#!/usr/bin/env python3
"""The checker (bench.py's cpu_baseline leg) at several thread counts on the GPU box's host: one 65 536-node cluster per thread is ~150 MB of
randomly accessed state, and the aggregate stops growing long before every hardware thread is busy.  usage: tools/cpu_baseline_sweep.py [threads ...]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("threads", type=int, nargs="*", default=[16, 32, 64, 128])
a = ap.parse_args()
for t in a.threads:
    args = argparse.Namespace(seed=1, nodes=65536, subject_cap=4, fanout=3, cpu_threads=t)
    r = bench.run_cpu_baseline(args, 2)
    print(json.dumps({"threads": r["cores"], "value": r["value"], "one_thread": r["one_thread"]["value"], "sample": r["sample"]}), flush=True)
