#!/bin/bash
# kernel resource usage of one .hip file (registers, scratch, LDS, occupancy): bash tools/kres.sh csrc/file.hip [extra flags]
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); print(); print(cur[:90],end=' ')
    for k in ('VGPRs:','AGPRs:','ScratchSize','Occupancy','LDS Size','SGPRs:','Spill'):
        m=re.search(k+r'[^:]*:? *(\d+)',l)
        if m and k in l: print(k.strip(':')[:7]+'='+m.group(1),end=' ')
print()
"
