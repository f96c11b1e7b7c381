#!/usr/bin/env python
"""Memory instructions, waits, branches and MFMAs of one kernel of a -save-temps gfx950 .s file, in program order
(to see whether a loop's loads are waited on with counted vmcnt or drained with vmcnt(0)):
    isa_loop_mem.py <file.s> <mangled-name-substring> [first-label last-label]"""
import re
import sys

txt = open(sys.argv[1]).read()
name = sys.argv[2]
i = txt.index(name)
i = txt.index('\n' + txt[txt.rindex('\n', 0, i) + 1:txt.index(':', i)] + ':', 0) if False else i
m = re.search(r'^(\S*' + re.escape(name) + r'\S*):', txt, re.M)
i = m.start()
j = txt.index('.Lfunc_end', i)
out = []
for l in txt[i:j].split('\n'):
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        out.append(m.group(1))
        continue
    op = s.split()[0] if s.split() else ''
    if (op.startswith('global_') or op.startswith('scratch_') or op.startswith('buffer_') or (op == 's_waitcnt' and 'vmcnt' in s)
            or op.startswith('s_cbranch') or op == 's_branch' or op.startswith('v_mfma')):
        out.append('   ' + s[:100])
if len(sys.argv) > 4:
    out = out[out.index(sys.argv[3]):out.index(sys.argv[4])]
print('\n'.join(out))
