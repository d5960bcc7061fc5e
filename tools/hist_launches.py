"""Diagnostic: per-launch durations of the network kernels in one headline step (rocprofv3 --kernel-trace of bench.py --steps 1):
the persistent trunk scales with the rows of a launch, so its duration histogram is the distribution of rows per lock-step."""
import csv, glob, sys, collections
f = glob.glob("gpurun_out/trace/**/k_kernel_trace.csv", recursive=True)[0]
tr, g1, g2 = [], [], []
for r in csv.DictReader(open(f)):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"]
    if "k_trunk2" in n: tr.append(d)
    elif "k_gemm_solo" in n:
        gs = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        (g1 if gs > 100000 else g2).append(d)
import numpy as np
tr = np.array(tr); g1 = np.array(g1); g2 = np.array(g2)
full = np.percentile(tr, 99)
print("trunk launches", len(tr), "p99 us", full, "fc1", len(g1), "fc2", len(g2))
# rows estimate = 32768 * trunk_time / full-time (persistent trunk scales with rows)
edges = [0, .1, .2, .3, .4, .5, .6, .7, .8, .85, .9, .95, 1.01]
frac = np.minimum(tr / full, 1.0)
h, _ = np.histogram(frac, edges)
tt = [tr[(frac >= a) & (frac < b)].sum() / 1e3 for a, b in zip(edges[:-1], edges[1:])]
print("rows/32768 bin : launches : trunk ms")
for (a, b), c, t in zip(zip(edges[:-1], edges[1:]), h, tt): print("%.2f-%.2f : %5d : %7.1f" % (a, b, c, t))
print("fc1 total ms %.1f  fc2 total ms %.1f" % (g1.sum() / 1e3, g2.sum() / 1e3))
for name, g in (("fc1", g1), ("fc2", g2)):
    hh, ee = np.histogram(g, [0, 20, 60, 100, 140, 180, 220, 240, 250, 260, 270, 300, 2000])
    print(name, "duration histogram (us):", list(zip([int(x) for x in ee[:-1]], hh)))
