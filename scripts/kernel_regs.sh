#!/bin/bash
# registers / scratch of every kernel of the library (the compiler's resource-usage remarks): name vgprs agprs scratch occupancy
cd "$(dirname "$0")/../pastml_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC -c pml_api.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
name = None; row = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        name = m.group(1); row = {}
        continue
    for key, pat in (('v', r' VGPRs: (\d+)'), ('a', r'AGPRs: (\d+)'), ('s', r'ScratchSize \[bytes/lane\]: (\d+)'), ('o', r'Occupancy \[waves/SIMD\]: (\d+)')):
        m = re.search(pat, line)
        if m: row[key] = m.group(1)
    if 'LDS Size' in line and name:
        print(name, row.get('v'), row.get('a'), row.get('s'), row.get('o'))
" | c++filt | sed 's/^void //; s/(.*) / /' | sort
