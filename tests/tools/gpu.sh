#!/bin/bash
# Local wrapper around gpurun: stamps the tree's git HEAD into .build_head (the GPU box gets a snapshot without .git; bench.py and
# tests/tools/pmc_summary.py read the stamp so that every PMC / rocprof summary names the build it was taken on).
# Usage: tests/tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/../.." || exit 1
{ git rev-parse HEAD 2>/dev/null | cut -c1-12; git diff --quiet HEAD -- quadraticprogramsolver_amd/csrc include 2>/dev/null || echo "+dirty"; } | tr -d '\n' > .build_head
exec /usr/local/graft/bin/gpurun "$@"
