#!/bin/bash
# kernel resource usage of one csrc/*.hip (VGPRs / spills / LDS per kernel): tools/kres.sh uce_gemm.hip
f=/root/repo/unified-concept-editing_amd/csrc/$1
[ -f "$1" ] && f=$(readlink -f "$1")      # (or any path: an older revision checked out to /tmp)
cd /tmp && /opt/rocm/bin/hipcc -I/root/repo/unified-concept-editing_amd/csrc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form $UCE_DEFINES \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().replace('(anonymous namespace)::','').split('(')[0][-70:]; vals={}
    for k in ('VGPRs','AGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','SGPRs','LDS Size [bytes/block]'):
        m=re.search(re.escape(k)+r': (\d+)',line)
        if m and ('Total' not in line or k=='SGPRs'): vals[k]=m.group(1)
    if 'LDS Size' in line and cur: print(cur, vals)
"
