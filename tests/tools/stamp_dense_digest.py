"""Adds `csrc_dense_sha16` (bench.py: the source digest without the sparse-only translation units) to the side-car metadata of committed summaries that were taken
before that field existed.  The digest is computed from the git revision the summary was measured on -- accepted only when that revision's WHOLE-tree digest equals the
`csrc_sha16` the summary already carries, so nothing is stamped onto a summary of another tree.
usage: python tests/tools/stamp_dense_digest.py <git-rev> <profiles glob prefix, e.g. r04_z>"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

rev, prefix = sys.argv[1], sys.argv[2]
ls = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "quadraticprogramsolver_amd/csrc/"], capture_output=True, text=True, check=True).stdout.split()
names = [x.split("/")[-1] for x in ls]
read = lambda rel: subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{os.path.normpath(os.path.join('quadraticprogramsolver_amd/csrc', rel))}"], capture_output=True, check=True).stdout
whole = bench.csrc_digest(read=read, listing=names)
dense = bench.csrc_digest(exclude=bench.SPARSE_ONLY_UNITS, read=read, listing=names)
print(f"{rev}: whole {whole} dense {dense}")
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", prefix + "*.meta.json")) + glob.glob(os.path.join(ROOT, "profiles", prefix + "*pmc_traffic_*.json"))):
    d = json.load(open(f))
    meta = d["_meta"] if "_meta" in d else d
    if meta.get("csrc_sha16") != whole:
        print(f"skip {os.path.basename(f)}: measured on {meta.get('csrc_sha16')}"); continue
    meta["csrc_dense_sha16"] = dense
    meta["csrc_dense_note"] = f"added afterwards by tests/tools/stamp_dense_digest.py from git {rev}, whose whole-tree digest equals csrc_sha16"
    json.dump(d, open(f, "w"), indent=1)
    print(f"stamped {os.path.basename(f)}")
