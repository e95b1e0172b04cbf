"""Ad-hoc: prints the end-of-round table of DESIGN.md section 6 from the committed bench lines (profiles/<tag>_bench_*.json).  Not a test.
usage: python tests/tools/design_table.py r04_z"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_z"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles")
def load(n):
    p = os.path.join(root, f"{tag}_bench_{n}.json")
    return json.load(open(p)) if os.path.exists(p) else None
def k(v): return f"{v / 1e3:.2f} k" if v >= 1e4 else f"{v:.0f}"
print("| config | value (1 MI355X) | dominant kernel (HIP events inside the timed region): avg, algorithmic bytes, frac of 8 TB/s, PMC traffic ÷ algorithmic | loop roofline | time-to-ε: GPU setup + loop / CPU port total, parity | CPU port, best of N (16 cores) |")
print("|---|---|---|---|---|---|")
for n in ("c2", "c2_trsv1024", "c5", "c3", "c4", "c1"):
    d = load(n)
    if not d: continue
    r = d.get("roofline") or {}; lr = d.get("loop_roofline") or {}; t = d.get("time_to_eps") or {}; c = d.get("cpu_baseline") or {}
    dom = f"`{r.get('kernel')}` {r.get('avg_launch_us')} µs, {r.get('algo_bytes_per_launch', 0) / 1e6:.1f} MB, **{r.get('frac')}**" + (f", {r.get('traffic_over_algorithmic')} ×" if r.get("traffic_over_algorithmic") else ", traffic n/a")
    loop = lr.get("frac_of_8TBs", "—")
    if t:
        cpu = t.get("cpu") or {}; par = t.get("parity") or {}
        tt = f"{t.get('ms_setup', '?')} + {t.get('ms_loop', '?')} ms" + (f" / {cpu.get('ms_total', 0) / 1e3:.2f} s" if cpu else "") + (f", parity {'ok' if par.get('ok') else par.get('ok')}" if par else "")
    else: tt = "—"
    cb = f"{k(c['value'])} {c.get('unit', '')} (min {k(c.get('value_min', c['value']))}, {c.get('samples', 1)} samples)" if c else "—"
    extra = ""
    if d.get("refactor"): extra = f"; refactor {d['refactor']['ms_each']} ms each = {round(100 * d['refactor']['share_of_loop_time'])} % of the loop"
    setup = f", setup {d['setup_ms']} ms" if d.get("setup_ms") is not None else ""
    print(f"| {n} | **{k(d['value'])} {d['unit']}**{setup}{extra} | {dom} | {loop} | {tt} | {cb} |")
