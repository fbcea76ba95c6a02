#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin or file)."""
import re, sys
s = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
for b in re.split(r'remark: [^\n]*Function Name: ', s)[1:]:
    name = b.split('\n')[0].split(' [')[0].strip()
    g = lambda k: (re.search(k + r': (\S+)', b) or [None, '?'])[1]
    print(name[:84].ljust(84), 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'SGPR', g('SGPRs'),
          'scratch', g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'), 'LDS', g(r'LDS Size \[bytes/block\]'))
