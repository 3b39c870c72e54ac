#!/bin/bash
# builds and runs the stand-alone probes on the GPU box; output -> gpurun_out/ubench.txt
cd "$(dirname "$0")/../.."
out=gpurun_out/ubench.txt; mkdir -p gpurun_out; : > $out
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w"
echo "== lat.hip (cycles; the stamp pair itself costs ~60)" >> $out
$HC tools/ubench/lat.hip -o /tmp/ub_lat && /tmp/ub_lat >> $out 2>&1
echo "== ldsbw.hip" >> $out
$HC tools/ubench/ldsbw.hip -o /tmp/ub_ldsbw && /tmp/ub_ldsbw >> $out 2>&1
echo "== potrf.hip (-DPK_STAMPS)" >> $out
$HC -mllvm -amdgpu-mfma-vgpr-form -DPK_STAMPS -I include -I unified-concept-editing_amd/csrc tools/ubench/potrf.hip -o /tmp/ub_potrf && /tmp/ub_potrf >> $out 2>&1
echo "== potrf.hip" >> $out
$HC -mllvm -amdgpu-mfma-vgpr-form -I include -I unified-concept-editing_amd/csrc tools/ubench/potrf.hip -o /tmp/ub_potrf && /tmp/ub_potrf >> $out 2>&1
cat $out
