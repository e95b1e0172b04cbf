"""The CPU-only randomised tools of tests/tools/ at a small case count, so that the regular CPU suite walks them (the long runs are under profiles/):
layout builders + interpreter against scipy, the symbolic L D L' analysis against an independent fill count, the C restatement against the numpy one."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(name, cases, seed):
    env = dict(os.environ); env.setdefault("OMP_NUM_THREADS", "2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", name), str(cases), str(seed)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("tool,cases,seed", [("cpu_fuzz_layouts.py", 40, 5), ("cpu_fuzz_ldl_symbolic.py", 60, 5), ("cpu_fuzz_csc.py", 25, 5)])
def test_layout_and_symbolic_tools_find_nothing(tool, cases, seed):
    text = run_tool(tool, cases, seed)
    assert re.search(rf"^{cases} cases, 0 bad", text, re.M), text[-1500:]
    assert text.count("\nok ") + text.startswith("ok ") == cases


def test_the_two_restatements_agree_at_fixed_rho():
    """Adaptive rho is a thresholded decision on residual ratios and is judged against a restatement's own sensitivity inside the tool (it names what it accepts and what it
    does not); with a fixed rho the C and the numpy restatement must agree outright."""
    text = run_tool("cpu_fuzz_oracles.py", 150, 5)
    fixed = [ln for ln in text.splitlines() if "adpt=False" in ln]
    assert len(fixed) >= 40
    assert not [ln for ln in fixed if ln.startswith(("MISMATCH", "ERROR"))], "\n".join(ln for ln in fixed if not ln.startswith("ok"))[:2000]
    assert not [ln for ln in text.splitlines() if ln.startswith("ERROR")]
