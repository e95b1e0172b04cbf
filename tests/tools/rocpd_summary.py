"""Per-kernel summary of a rocprofv3 rocpd database (the default output format): calls, average and total duration per (kernel, grid).
usage: python tests/tools/rocpd_summary.py gpurun_out/prof/x_results.db [rows]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = db.execute("select name, count(*), avg(end-start), sum(end-start), grid_x, grid_y, workgroup_x from kernels "
                  "group by name, grid_x, grid_y order by sum(end-start) desc limit ?", (n,)).fetchall()
for r in rows:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r[0]); nm = m.group(1) if m else r[0][:40]
    print(f"{nm:44s} calls {r[1]:6d} avg_us {r[2]/1e3:8.2f} total_ms {r[3]/1e6:8.1f} grid {r[4] // max(r[6], 1)}x{r[5]} wg {r[6]}")
