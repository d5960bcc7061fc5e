#!/usr/bin/env python3
"""Condenses what tools/refresh_profiles.sh wrote under gpurun_out/ into the small files kept under profiles/:
  TAG_kernel_stats.csv    rocprofv3 --stats of the bench command
  TAG_traffic.json        HBM bytes per full-batch launch of every network kernel, per workload (FETCH_SIZE / WRITE_SIZE passes)
  TAG_mfma_counters.json  SQ_VALU_MFMA_BUSY_CYCLES etc. per network kernel, per workload
  TAG_kstep_counters.json k_step<true,true> inside self-play: duration, HBM bytes, wait shares
  TAG_counters.csv        every (workload, kernel, counter) average the json files were built from"""
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
WORKLOADS = [("othello", 32768), ("othello", 4096), ("connect4", 8192)]
FC1_KT = {"othello": "16", "connect4": "6"}  # k_gemm's last template argument KT = K / 32 of fc1 (fc2: 32 / 2)


def find(d, suffix):
    hits = glob.glob(os.path.join(out, d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


for d, name in ((f"{tag}_prof", f"{tag}_kernel_stats.csv"), (f"{tag}_prof_c2", f"{tag}_c2_kernel_stats.csv"), (f"{tag}_prof_c4", f"{tag}_c4_kernel_stats.csv")):
    stats = find(d, "kernel_stats.csv")
    if stats:
        open(os.path.join(out, name), "w").write(open(stats).read())


SOLO_GRIDS = set()


def classify(name, game, grid=None):
    if "k_gemm_solo" in name:  # one template for both dense layers: fc1 (N = 1024) launches twice the workgroups of fc2 (N = 512)
        SOLO_GRIDS.add(int(grid or 0))
        return ("k_gemm_fc1", "k_gemm_fc2", int(grid or 0))
    if "k_trunk" in name:
        return "k_trunk"
    if "k_heads" in name:
        return "k_heads"
    if "k_tail_small" in name or "k_tail_mfma" in name:
        return "k_tail"
    if "k_gemm" in name or "k_dense" in name:
        kt = re.search(r"k_gemm<[^>]*,\s*(\d+)>", name)
        return "k_gemm_fc1" if kt and kt.group(1) == FC1_KT[game] else "k_gemm_fc2"
    if re.search(r"k_step<\s*true,\s*true\s*>", name) or "k_stepILb1ELb1" in name:
        return "k_step"
    return None


def averages(d, game):
    """{(kernel class, counter): (mean value, mean duration ns, dispatches)} of one counter pass"""
    f = find(d, "counter_collection.csv")
    acc, pending = {}, []
    if not f:
        return acc
    SOLO_GRIDS.clear()
    for row in csv.DictReader(open(f)):
        k = classify(row["Kernel_Name"], game, row.get("Grid_Size"))
        if k is None:
            continue
        if isinstance(k, tuple):
            pending.append((k, row))
            continue
        dur = None
        if row.get("Start_Timestamp") and row.get("End_Timestamp"):
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        a = acc.setdefault((k, row["Counter_Name"]), [0.0, 0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += dur or 0.0; a[2] += 1
    big = max(SOLO_GRIDS) if SOLO_GRIDS else 0
    for (k1, k2, grid), row in pending:  # k_gemm_solo: the larger grid is fc1 (when only one grid size ran solo it is fc1: fc2 needs 32768 rows)
        k = k1 if grid == big else k2
        dur = None
        if row.get("Start_Timestamp") and row.get("End_Timestamp"):
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        a = acc.setdefault((k, row["Counter_Name"]), [0.0, 0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += dur or 0.0; a[2] += 1
    return {k: (v[0] / v[2], v[1] / v[2], v[2]) for k, v in acc.items()}


rows, traffic, mfma, kstep = [], {}, {}, {}
for game, B in WORKLOADS:
    key = f"{game}_{B}"
    fetch, write = averages(f"{tag}_pmc_fetch_{game}_{B}", game), averages(f"{tag}_pmc_write_{game}_{B}", game)
    t = {}
    for k in ("k_trunk", "k_gemm_fc1", "k_gemm_fc2", "k_heads", "k_tail"):
        if (k, "FETCH_SIZE") in fetch and (k, "WRITE_SIZE") in write:
            f_kb, w_kb = fetch[(k, "FETCH_SIZE")][0], write[(k, "WRITE_SIZE")][0]
            t[k] = int((2 * f_kb + w_kb) * 1024)  # gfx950: FETCH_SIZE counts half of a wide read (MI355X_MICROARCH.md, HBM)
            rows.append((key, k, "FETCH_SIZE_KB", f_kb, fetch[(k, "FETCH_SIZE")][2])); rows.append((key, k, "WRITE_SIZE_KB", w_kb, write[(k, "WRITE_SIZE")][2]))
    if t:
        traffic[key] = t
    m = {}
    for d in (f"{tag}_pmc_mfma_{game}_{B}", f"{tag}_pmc_mops_{game}_{B}"):
        for (k, c), (v, dur, n) in averages(d, game).items():
            if k == "k_step":
                continue
            m.setdefault(k, {})[c] = v
            if dur:
                m[k]["duration_us_under_pmc"] = dur / 1e3
            rows.append((key, k, c, v, n))
    for k, c in m.items():
        # share of the chip's SIMD-cycles in which the matrix pipe was busy.  rocprofv3 sums a counter over its instances:
        # SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cross-check: SQ_INSTS_MFMA x 64 cycles per 32x32x2 f32 MFMA),
        # GRBM_GUI_ACTIVE over the 8 XCDs (cross-check: / 8 / kernel duration = 2.1-2.4 GHz) -> busy / (active / 8 * 1024)
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0)
            if c.get("duration_us_under_pmc"):
                c["clock_ghz_from_gui_active"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (c["duration_us_under_pmc"] * 1e3)
    if m:
        mfma[key] = m
    ks = {}
    for d, ctrs in ((f"{tag}_kstep_fetch_{game}_{B}", ["FETCH_SIZE"]), (f"{tag}_kstep_write_{game}_{B}", ["WRITE_SIZE"]),
                    (f"{tag}_kstep_sq_{game}_{B}", None)):
        for (k, c), (v, dur, n) in averages(d, game).items():
            if k != "k_step":
                continue
            ks[c] = v
            ks["launches_averaged"] = n
            if dur:
                ks["duration_us_under_pmc"] = dur / 1e3
            rows.append((key, k, c, v, n))
    if "FETCH_SIZE" in ks and "WRITE_SIZE" in ks:
        ks["hbm_bytes_per_launch_upper"] = int((2 * ks["FETCH_SIZE"] + ks["WRITE_SIZE"]) * 1024)  # x2: calibrated for wide streams only; 16/32-byte node accesses are uncalibrated
        ks["hbm_bytes_per_launch_lower"] = int((ks["FETCH_SIZE"] + ks["WRITE_SIZE"]) * 1024)
        ks["algorithmic_bytes_per_launch"] = 919 * B
    if ks.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in ks:
                ks[c + "_share_of_wave_cycles"] = ks[c] / ks["SQ_WAVE_CYCLES"]
    if ks:
        kstep[key] = ks
sys.path.insert(0, root)
from alphazero_amd._lib import csrc_tree_hash  # noqa: E402

# the hash of the sources the measured library was built from: bench.py uses a counter file only for the same tree
for d, comp in ((traffic, "net"), (mfma, "net"), (kstep, "engine")):
    if d:
        d["csrc_component"] = comp  # the network kernels' counters stay valid while only the engine or the training step changes
        d["csrc_sha"] = csrc_tree_hash(comp)
if traffic:
    json.dump(traffic, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
if mfma:
    json.dump(mfma, open(os.path.join(out, f"{tag}_mfma_counters.json"), "w"), indent=1)
if kstep:
    json.dump(kstep, open(os.path.join(out, f"{tag}_kstep_counters.json"), "w"), indent=1)
if rows:
    with open(os.path.join(out, f"{tag}_counters.csv"), "w") as fh:
        fh.write("workload,kernel,counter,mean_per_launch,launches\n")
        for r in rows:
            fh.write(",".join(str(x) for x in r) + "\n")
print("collected", tag, "csrc", csrc_tree_hash(), "traffic", list(traffic), "mfma", list(mfma), "kstep", list(kstep))
