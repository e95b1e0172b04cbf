"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced
streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Usage: pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv [out.json] [command that was profiled]
The summary carries `_meta`: the git HEAD and the digest of the kernel sources it was measured on (bench.py reports a traffic figure only
when that digest is the running build's)."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_head, csrc_digest, SPARSE_ONLY_UNITS


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        mm = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"]); name = mm.group(1) if mm else r["Kernel_Name"]
        d[name].append(float(r["Counter_Value"]))
    return d


f, w = load(sys.argv[1]), load(sys.argv[2])
out = {"_meta": {"head": build_head(), "csrc_sha16": csrc_digest(), "csrc_dense_sha16": csrc_digest(exclude=SPARSE_ONLY_UNITS), "command": sys.argv[4] if len(sys.argv) > 4 else None,
                 "method": "two separate rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, --kernel-trace only); hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE"}}
for k in sorted(f, key=lambda k: -sum(f[k])):
    if not k.startswith("k_"): continue
    fetch_kib = sum(f[k]) / len(f[k]); write_kib = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    out[k] = {"launches": len(f[k]), "FETCH_SIZE_KiB_raw_per_launch": round(fetch_kib, 1), "WRITE_SIZE_KiB_per_launch": round(write_kib, 1),
              "hbm_read_bytes_corrected_x2": round(fetch_kib * 1024 * 2), "hbm_write_bytes": round(write_kib * 1024),
              "hbm_bytes_per_launch": round(fetch_kib * 1024 * 2 + write_kib * 1024)}
    print(f"{k:42s} launches {len(f[k]):5d} fetch(raw) {fetch_kib/1024:9.2f} MiB  write {write_kib/1024:8.2f} MiB  -> traffic (2x fetch + write) {out[k]['hbm_bytes_per_launch']/1e6:9.2f} MB")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
