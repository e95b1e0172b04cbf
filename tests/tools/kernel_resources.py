"""Print VGPR / spill / scratch / LDS figures of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tests/tools/kernel_resources.py quadraticprogramsolver_amd/csrc/k_pass.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-I", os.path.join(root, "include"),
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]}
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "VGPR Spill", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key.split(" ")[0]] = int(m.group(1))
            if key.startswith("LDS"):
                if flt in cur["name"]:
                    print(f'{cur["name"][:90]:90s} vgpr {cur.get("VGPRs")} agpr {cur.get("AGPRs")} scratch {cur.get("ScratchSize")} occ {cur.get("Occupancy")} lds {cur.get("LDS")}')
                cur = None
