#!/usr/bin/env python
"""Instruction mix of the largest loop of one kernel (static count from the gfx950 ISA):
    python tools/isa_loop_mix.py vit-adapter_amd/csrc/attn_fwd.hip attn_fwd_kernel
VALU-bound kernels (attention at head_dim 64) are tuned by this count: cycles ~ 4 per VALU instruction, 16 per
transcendental, 32 per v_mfma_f32_32x32x16_bf16, per wave, with the waves of a SIMD sharing each pipe."""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_check import HIPCC, ROOT, per_file_flags  # noqa: E402


def main():
    src, name = os.path.abspath(sys.argv[1]), sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-I', os.path.join(ROOT, 'include'),
               '-I', os.path.dirname(src), '-c', src, '-o', os.path.join(tmp, 'x.o'), '-save-temps'] + per_file_flags(src)
        subprocess.check_call(cmd, cwd=tmp, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')]
        text = open(os.path.join(tmp, asm[0])).read()
    for m in re.finditer(r'^(_Z\S*%s\S*):.*?\n(.*?)^\.Lfunc_end' % re.escape(name), text, re.S | re.M):
        lines = m.group(2).split('\n')
        labels = {l.strip()[:-1]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l.strip())}
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r's_cbranch\S*\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                loops.append((labels[mm.group(1)], i))
        spans = sorted(loops, key=lambda s: s[0] - s[1])[:int(os.environ.get('LOOPS', '1'))] or [(0, len(lines))]
        for a, b in spans:
            c = collections.Counter()
            for l in lines[a:b]:
                l = l.strip()
                if not l or l[0] in '.;':
                    continue
                op = l.split()[0]
                if op.startswith('s_') and not op.startswith(('s_waitcnt', 's_barrier')):
                    op = 'salu'
                c[op] += 1
            valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
            trans = sum(v for k, v in c.items() if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', k))
            mfma = sum(v for k, v in c.items() if 'mfma' in k)
            print('%s lines %d-%d: %d instructions, %d VALU (%d transcendental), %d MFMA -> ~%d VALU cycles vs %d MFMA cycles per wave'
                  % (m.group(1)[:70], a, b, sum(c.values()), valu, trans, mfma, 4 * valu + 12 * trans, 32 * mfma))
            for k, v in c.most_common(int(os.environ.get('TOP', '24'))):
                print('   %-28s %d' % (k, v))


if __name__ == '__main__':
    main()
