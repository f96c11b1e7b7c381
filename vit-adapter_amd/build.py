#!/usr/bin/env python
"""Build libvitadapter_hip.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box).

    python vit-adapter_amd/build.py [--force] [--verbose]

One object per csrc/*.hip (recompiled only when the source or a header is newer), linked into
vit-adapter_amd/lib/libvitadapter_hip.so.  No torch headers are involved: the library's only
interface is the C ABI in include/vitadapter_hip.h.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libvitadapter_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wall', '-Wno-unused-function',
         '-ffp-contract=fast']


# The attention kernels keep MFMA accumulators that the VALU works on between matrix steps (softmax, dS): with the
# compiler's default the accumulators live in AGPRs and every such step pays a v_accvgpr_read/write pair per register
# (190 of 930 instructions in the forward's key loop) and the kernels need 212 - 288 registers (1 - 2 waves per SIMD).
# Forcing the VGPR form of MFMA removes the copies and brings them to 146 - 202 registers.
PER_FILE_FLAGS = {f: ['-mllvm', '-amdgpu-mfma-vgpr-form'] for f in ('attn_fwd.hip', 'attn_bwd.hip', 'attn_win.hip', 'attn_flash.hip')}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


EXTRA_FLAGS = os.environ.get('VAH_EXTRA_HIPCC_FLAGS', '').split()       # experiments (tools/ablate_tile.sh): -D switches


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(
        os.path.join(HERE, '..', 'include', '*.h'))
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(s), []) + EXTRA_FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _newer(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [
            '-L/opt/rocm/lib', '-lhipblaslt']       # resolved at run time by the copy torch has loaded
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or True))
