"""rocprofv3 kernel_trace.csv -> a per-kernel summary in the column layout of rocprofv3's own kernel_stats.csv, with two corrections the --stats summary cannot make
(round-3 review item 7):
  * a kernel name launched with several grids is split into one row per grid (name + " [grid X x Y x Z]", grid in workgroups): the two column-blocked products of a CG
    iteration -- [P; A] u and A' v -- run the same kernel and --stats reports their joint average;
  * dispatches of the CG kernels that returned at their first instruction (the `done` flag of an iteration enqueued past convergence: k_sparse.hip) are counted in a row
    of their own (name + " [returned at the done flag]") instead of dragging the average down: a dispatch of k_spmv_* / k_cg_* shorter than 3 us did no work.
usage: trace_stats_split.py <kernel_trace.csv> <out_stats.csv>"""
import collections
import csv
import math
import re
import sys

NOOP_NS = 3000
EARLY_EXIT = re.compile(r"k_spmv_(sell|blk|stream|combine)<|k_cg_")

rows = csv.DictReader(open(sys.argv[1]))
groups = collections.defaultdict(list)
grids = collections.defaultdict(set)
for r in rows:
    name = r["Kernel_Name"]
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    wg = [max(1, int(r.get(f"Workgroup_Size_{a}", 1) or 1)) for a in "XYZ"]
    g = tuple(max(1, int(r.get(f"Grid_Size_{a}", 1) or 1)) // w for a, w in zip("XYZ", wg))
    if EARLY_EXIT.search(name) and dur < NOOP_NS:
        groups[(name, None)].append(dur)
        continue
    groups[(name, g)].append(dur)
    grids[name].add(g)
total = sum(sum(v) for v in groups.values()) or 1
out = []
for (name, g), d in groups.items():
    label = name
    if g is None:
        label += " [returned at the done flag]"
    elif len(grids[name]) > 1:
        label += " [grid %d x %d x %d]" % g
    n = len(d); s = sum(d); avg = s / n
    sd = math.sqrt(sum((x - avg) ** 2 for x in d) / n)
    out.append((label, n, s, avg, 100.0 * s / total, min(d), max(d), sd))
out.sort(key=lambda t: -t[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for t in out:
        w.writerow([t[0], t[1], t[2], round(t[3], 6), round(t[4], 2), t[5], t[6], round(t[7], 6)])
print(f"{sum(len(v) for v in groups.values())} dispatches, {len(out)} rows, {sum(len(v) for (n, g), v in groups.items() if g is None)} returned at the done flag")
