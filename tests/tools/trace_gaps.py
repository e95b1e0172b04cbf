"""Summarise a rocprofv3 kernel_trace.csv: per-kernel average duration and the idle gaps between consecutive kernels."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -2000:]
dur = collections.defaultdict(list); gap_after = collections.defaultdict(list)
for a, b in zip(tail[:-1], tail[1:]):
    mm = re.search(r"(k_\w+(<[^>]*>)?)", a["Kernel_Name"]); n = (mm.group(1) if mm else a["Kernel_Name"])[:42]
    dur[n].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    gap_after[n].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
tot = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
print(f"window {tot:.0f} us, {len(tail)} kernels")
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[n], gap_after[n]
    print(f"{n:42s} calls {len(d):5d} avg {sum(d)/len(d):8.2f} us  sum {sum(d):9.0f} us ({100*sum(d)/tot:4.1f}%)  gap-after avg {sum(g)/len(g):6.2f} us")
print(f"total busy {sum(sum(v) for v in dur.values()):.0f} us, idle {sum(sum(v) for v in gap_after.values()):.0f} us")
