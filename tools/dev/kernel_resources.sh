#!/bin/bash
# Developer script (no GPU needed): registers / spills / scratch / occupancy of every kernel in kernels.hip, from hipcc's
# -Rpass-analysis=kernel-resource-usage remarks.  usage: tools/dev/kernel_resources.sh [grep pattern]
cd "$(dirname "$0")/../../iwae_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c kernels.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = {'name': m.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for key, pat in (('v', r' VGPRs: *(\d+)'), ('a', r'AGPRs: *(\d+)'), ('scr', r'ScratchSize \[bytes/lane\]: *(\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: *(\d+)'),
                     ('spill', r'VGPRs Spill: *(\d+)'), ('lds', r'LDS Size \[bytes/block\]: *(\d+)'), ('s', r' SGPRs: *(\d+)')):
        m = re.search(pat, l)
        if m: cur[key] = m.group(1)
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    print('%-88s vgpr %3s agpr %3s sgpr %3s spill %2s scratch %3s occ %s' % (n[:88], r.get('v'), r.get('a'), r.get('s'), r.get('spill'), r.get('scr'), r.get('occ')))
" | grep -E "${1:-.}"
