#!/bin/bash
# usage: tools/resource_usage.sh [CORA_TU] [CORA_LDG] [name filter]  -- registers / scratch / LDS / occupancy of the kernels of
# one translation unit of kernels.hip as the compiler reports them (no GPU needed); RES_FLAGS="-D..." adds compiler flags
cd "$(dirname "$0")/.." || exit 1
TU=${1:-1}; LDG=${2:-1}; FILTER=${3:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Icora_amd/csrc -DCORA_TU=$TU -DCORA_LDG=$LDG $RES_FLAGS -x hip \
  -c cora_amd/csrc/kernels.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import re, subprocess, sys
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        if cur: rows.append(cur)
        cur = {'name': v}
    else:
        cur[k.split(' ')[0]] = v
if cur: rows.append(cur)
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
print('%-52s %5s %5s %8s %6s %5s' % ('kernel', 'VGPR', 'AGPR', 'scratch', 'LDS', 'occ'))
for r, n in zip(rows, names):
    n = n.replace('void cora::', '').split('(')[0]
    if re.search(sys.argv[1], n):
        print('%-52s %5s %5s %8s %6s %5s' % (n[:52], r.get('VGPRs'), r.get('AGPRs', '0'), r.get('ScratchSize'), r.get('LDS'), r.get('Occupancy')))
" "$FILTER"
