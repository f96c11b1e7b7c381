#!/usr/bin/env python
"""Registers / scratch / occupancy per kernel of one csrc/*.hip file (no GPU needed): kres.py <file.hip> [substring]"""
import os
import re
import subprocess
import sys

src = os.path.abspath(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ''
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from isa_check import per_file_flags  # noqa: E402

cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=fast', '-c', src, '-o', '/dev/null',
       '-Rpass-analysis=kernel-resource-usage'] + per_file_flags(src)
txt = subprocess.run(cmd, capture_output=True, text=True, cwd='/tmp').stderr
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(anonymous namespace\)::|vah::', '', dn)
    dn = re.sub(r'\(.*$', '', dn)[:110]
    if pat in dn:
        print('%-110s sgpr %3s vgpr %3s agpr %3s scratch %4s occ %s lds %s' % (
            dn, g('TotalSGPRs'), g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
            g(r'LDS Size \[bytes/block\]')))
