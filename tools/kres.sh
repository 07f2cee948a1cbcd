#!/bin/bash
# usage: tools/kres.sh [pattern] — registers, scratch and occupancy of the v2 kernels (compiles dcrx_kernels_v2.hip)
cd /root/repo/decombinator_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c dcrx_kernels_v2.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "${@:2}" 2>&1 | python3 -c "
import sys,re
pat=sys.argv[1] if len(sys.argv)>1 else 'ILb1ELi10E'
cur=None; rows={}
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([\w\[\]/ ]+): (\d+)',ln)
    if m and cur: rows[cur][m.group(1).strip()]=int(m.group(2))
for k,v in rows.items():
    if pat in k: print(k[9:44], 'VGPR',v.get('VGPRs'),'spill',v.get('VGPRs Spill'),'SGPRspill',v.get('SGPRs Spill'),'scratch',v.get('ScratchSize [bytes/lane]'),'occ',v.get('Occupancy [waves/SIMD]'))
" "${1:-ILb1ELi10E}"
