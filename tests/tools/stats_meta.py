"""Side-car of a rocprofv3 --kernel-trace --stats summary: which build and which command it was taken on.
Usage: stats_meta.py <kernel_stats.csv> "<command>"  -> writes <kernel_stats>.meta.json (bench.py shows rocprof averages beside its HIP-event
averages only when the digest matches the running build)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_head, csrc_digest, SPARSE_ONLY_UNITS
json.dump({"head": build_head(), "csrc_sha16": csrc_digest(), "csrc_dense_sha16": csrc_digest(exclude=SPARSE_ONLY_UNITS), "command": sys.argv[2] if len(sys.argv) > 2 else None}, open(sys.argv[1][:-4] + ".meta.json", "w"), indent=1)
