#!/usr/bin/env python3
"""Histogram of the trunk kernel's launch durations in a rocprofv3 kernel trace (the persistent trunk's time is quantised in rounds
of 2048 boards, so the histogram is the distribution of rounds per lock-step), and the mean duration by position inside a ply.
  rocprofv3 --kernel-trace --output-format csv -d DIR -o k -- python3 tools/run_config.py connect4 8192 4
  python tools/trunk_round_hist.py DIR [sims]"""
import csv
import glob
import sys
from collections import Counter

d = sys.argv[1]
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 200
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
trunk = [(s, e) for s, e, n in rows if "k_trunk" in n]
dur = [(e - s) / 1e3 for s, e in trunk]
print("trunk launches", len(dur), "mean us", sum(dur) / len(dur))
h = Counter(int(x // 4) * 4 for x in dur)
for k in sorted(h):
    print(f"{k:4d}-{k + 4:<4d} us {h[k]:6d} {'#' * (60 * h[k] // max(h.values()))}")
# position inside a ply: the searches of a ply are sims + 1 network calls (root prior + sims); k_move separates the plies
ply, pos, acc = 0, 0, {}
it = iter(rows)
for s, e, n in rows:
    if n.startswith("k_move"):
        pos = 0
        continue
    if "k_trunk" in n:
        acc.setdefault(pos, []).append((e - s) / 1e3)
        pos += 1
print("mean trunk us by position inside a ply (every 10th):")
for p in sorted(acc):
    if p % 10 == 0 or p < 4:
        v = acc[p]
        print(f"  pos {p:3d}: {sum(v) / len(v):6.1f} us over {len(v)} plies")
