#!/usr/bin/env python3
"""Condenses what tools/refresh_profiles.sh wrote under gpurun_out/ into the small files kept under profiles/."""
import csv
import glob
import json
import os
import re
import sys

tag, G = sys.argv[1], int(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")


def find(d, suffix):
    hits = glob.glob(os.path.join(out, d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


stats = find(f"{tag}_prof", "kernel_stats.csv")
if stats:
    open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w").write(open(stats).read())

def per_kernel(d, counter):
    f = find(d, "counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        if "k_trunk" in name:
            k = "k_trunk"
        elif "k_heads" in name:
            k = "k_heads"
        elif "k_gemm" in name:
            kt = re.search(r"k_gemm<[^>]*,\s*(\d+)>", name)  # last template argument KT = K / 32: 16 -> fc1 (K=512), 32 -> fc2
            k = "k_gemm_fc1" if kt and kt.group(1) == "16" else "k_gemm_fc2"
        else:
            continue
        acc.setdefault(k, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(f"{tag}_pmc_fetch", "FETCH_SIZE")
write = per_kernel(f"{tag}_pmc_write", "WRITE_SIZE")
if fetch and write:
    traffic = {"batch": G}
    with open(os.path.join(out, f"{tag}_hbm_counters.csv"), "w") as fh:
        fh.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
        for k in ("k_trunk", "k_gemm_fc1", "k_gemm_fc2", "k_heads"):
            if k in fetch and k in write:
                b = int((2 * fetch[k] + write[k]) * 1024)  # gfx950: FETCH_SIZE counts half of a wide read (MI355X_MICROARCH.md, HBM)
                traffic[k] = b
                fh.write(f"{k},{fetch[k]:.1f},{write[k]:.1f},{b}\n")
    json.dump(traffic, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
print("collected", tag)
