import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:n]:
    mm = re.search(r"(k_\w+(<[^>]*>)?)", r["Name"]); name = (mm.group(1) if mm else r["Name"])[:48]
    tagm = re.search(r"(\[(grid|returned)[^\]]*\])\s*$", r["Name"])
    if tagm: name = (name[:28] + " " + tagm.group(1).replace("returned at the done flag", "no-op"))[:48]
    print(f"{name:48s} calls {str(r['Calls']):>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {float(r['Percentage']):5.1f}")
