#!/usr/bin/env python3
"""tools/emu/race_report.py REPORT [LIBRARY] — symbolise what tools/emu/emu_race.cpp wrote (EMU_RACE_OUT): one line per pair of source
positions (innermost frame inside consul_amd/csrc of either access), occurrences summed, most frequent first."""
import collections, re, subprocess, sys
rep = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else "tools/emu/_build/libswimsim_emu_race.so"
rows = []
for line in open(rep):
    if line.startswith("#") or not line.strip():
        continue
    a, b, n, rest = line.split(" ", 3)
    rows.append((a, b, int(n), rest.split("|")[0].strip()))
addrs = sorted({r[0] for r in rows} | {r[1] for r in rows})
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--inlines", "-e", lib] + addrs, capture_output=True, text=True).stdout
where = {}
for addr, block in zip(addrs, out.strip().split("\n\n")):
    ls = block.strip().split("\n")
    frames = [(ls[i], ls[i + 1]) for i in range(0, len(ls) - 1, 2)]
    pos = [(fn, re.sub(r".*/csrc/", "", loc)) for fn, loc in frames if "/csrc/" in loc and ":0:" not in loc]
    inner = pos[0] if pos else (frames[0] if frames else ("?", "?"))
    outer = pos[-1] if pos else inner
    where[addr] = f"{inner[1]} [{re.sub(r'<.*', '', outer[0].split('(')[0])}]"
agg = collections.Counter()
for a, b, n, kind in rows:
    agg[(kind, where[a], where[b])] += n
for (kind, wa, wb), n in agg.most_common():
    print(f"{n:>10}  {kind:<28} {wa}  <->  {wb}")
