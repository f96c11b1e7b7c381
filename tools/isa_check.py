#!/usr/bin/env python
"""Static check of the gfx950 ISA of one csrc/*.hip file (no GPU needed): registers, spills, LDS and -
the finding that paid most in round 1 - global loads that are waited on one by one.

A load placed under a condition (`if (valid) v = p[i]`, `ptr ? *ptr : 0`) is compiled into its own branch
with its own `s_waitcnt vmcnt(0)`: N such loads cost N memory round trips instead of one.  The script
counts `global_load` / `buffer_load` instructions that are directly followed (among memory instructions)
by `s_waitcnt vmcnt(0)`; more than one or two per kernel deserves a look at the source.

    python tools/isa_check.py vit-adapter_amd/csrc/msda_fused.hip [name substring ...] [-D...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def per_file_flags(src):
    """The extra flags vit-adapter_amd/build.py compiles this file with."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('vah_build', os.path.join(ROOT, 'vit-adapter_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.PER_FILE_FLAGS.get(os.path.basename(src), [])


def main():
    src = os.path.abspath(sys.argv[1])
    pats = [a for a in sys.argv[2:] if not a.startswith('-D')]
    defs = [a for a in sys.argv[2:] if a.startswith('-D')]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-I', os.path.join(ROOT, 'include'),
               '-I', os.path.dirname(src), '-c', src, '-o', os.path.join(tmp, 'x.o'), '-save-temps'] + defs + per_file_flags(src)
        subprocess.check_call(cmd, cwd=tmp, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')]
        text = open(os.path.join(tmp, asm[0])).read()
    meta = {}
    rx = (r'- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?'
          r'\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)')
    for m in re.finditer(rx, text, re.S):
        meta[m.group(3)] = (int(m.group(4)), int(m.group(1)), int(m.group(5)), int(m.group(2)))
    print('%-78s %5s %5s %5s %6s %5s %6s' % ('kernel', 'vgpr', 'agpr', 'spill', 'lds', 'loads', 'serial'))
    for m in re.finditer(r'^(_Z\S+):\s.*?^\.Lfunc_end', text, re.S | re.M):
        name = m.group(1)
        if name not in meta or (pats and not any(p in name for p in pats)):
            continue
        ev = [ln for ln in m.group(0).split('\n') if 'global_load' in ln or 'buffer_load' in ln or 's_waitcnt vmcnt' in ln]
        loads = sum('_load' in e for e in ev)
        serial = sum('_load' in a and 'vmcnt(0)' in b for a, b in zip(ev, ev[1:]))
        v, a, sp, lds = meta[name]
        demangled = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
        short = demangled.replace('(anonymous namespace)::', '').replace('void ', '')
        short = re.sub(r'\(.*', '', short) if short != name else name[name.find('N_1') + 3:]
        print('%-78s %5d %5d %5d %6d %5d %6d' % (short[:78], v, a, sp, lds, loads, serial))


if __name__ == '__main__':
    main()
