#!/bin/bash
# builds the ad-hoc micro-benchmarks next to their sources (gfx950 only)
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../../.." && pwd)"
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-value -Wno-unused-result -I"$root/quadraticprogramsolver_amd/csrc" -I"$root/include" "$here/$f.hip" -o "$here/$f" 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|In file included" || true
  ls -la "$here/$f"
done
