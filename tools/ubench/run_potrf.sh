#!/bin/bash
# usage: run_potrf.sh "<defines>" ...   (each argument = one variant's -D flags; "" = the product form)
cd "$(dirname "$0")/../.."
for defs in "$@"; do
  echo "== variant: ${defs:-product}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -mllvm -amdgpu-mfma-vgpr-form $defs -I include -I unified-concept-editing_amd/csrc -I tools/ubench \
      tools/ubench/potrf.hip -o /tmp/potrf_v && /tmp/potrf_v
done
