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
echo "== gemm_w1.hip (one wave per SIMD: BN = 256 and BN = 320, BK = 64; accumulators in AGPRs: no -amdgpu-mfma-vgpr-form)" >> $out
for bn in 256 320; do $HC -DBNV=$bn -DBKV=64 -DNSTV=2 tools/ubench/gemm_w1.hip -o /tmp/ub_gemm_w1 && /tmp/ub_gemm_w1 >> $out 2>&1; done
cat $out
