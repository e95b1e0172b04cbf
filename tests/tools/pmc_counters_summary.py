"""Per-kernel means of arbitrary rocprofv3 --pmc counters.  Usage: pmc_counters_summary.py out.json pass1_counter_collection.csv [pass2.csv ...]
Every pass is a separate run of the same command (the counters of one pass share the hardware slots); values are averaged over the
dispatches of a kernel name, k_spmv_blk additionally split by its grid (the two products of a CG iteration run the same instantiation)."""
import collections
import csv
import json
import re
import sys

out = collections.defaultdict(dict)
for path in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        mm = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        if not mm:
            continue
        name = mm.group(1)
        if name.startswith("k_spmv_blk"):
            name += f" grid={r.get('Grid_Size', '?')}"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        for c, v in cs.items():
            out[name][c] = round(sum(v) / len(v), 2)
            out[name]["dispatches"] = len(v)
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for name in sorted(out, key=lambda k: -out[k].get("dispatches", 0))[:8]:
    print(name, json.dumps(out[name], sort_keys=True))
