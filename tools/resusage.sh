#!/bin/bash
# usage: resusage.sh <hipcc args...>   -- compile and print one line per kernel: name vgpr agpr scratch occupancy
/opt/rocm/bin/hipcc "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)',l)
    if not m:
        if 'error' in l: print(l,end='')
        continue
    k,v=m.groups()
    if k=='Function Name':
        if cur: print(cur)
        cur=v
    else: cur+=f' {k.split()[0]}={v}'
if cur: print(cur)
"
