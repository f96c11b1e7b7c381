#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes of tools/prof_msda_single.py into profiles/r03_msda_pmc.json.

Layout expected (one directory per pass, csv output):
    <root>/<cfg>_n<noise>_<COUNTER>/**/*counter_collection.csv      COUNTER in FETCH_SIZE, WRITE_SIZE
    python tools/pmc_msda_summary.py <root> <iters> > profiles/r03_msda_pmc.json
Counter unit KiB.  hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950: FETCH_SIZE reports half of
the bytes of wide coalesced reads, MI355X_MICROARCH.md "HBM"); per call = sum over the kernels of one
forward (msda_win_schedule + msda_fused_fwd_win, or msda_fused_fwd) / backward (msda_plan + msda_bin + msda_tile +
msda_grad_finish) call.  The JSON is stamped with the digest of the MSDA kernel
sources (bench.py::msda_source_digest): bench.py reports it as roofline.traffic only while they are unchanged.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root, iters = sys.argv[1], int(sys.argv[2])
    acc = defaultdict(lambda: defaultdict(float))          # (cfg_noise, dir) -> counter -> KiB per call
    for d in sorted(glob.glob(os.path.join(root, '*_*SIZE'))):
        base = os.path.basename(d)
        counter = 'FETCH_SIZE' if base.endswith('FETCH_SIZE') else 'WRITE_SIZE'
        key = base[:-len(counter) - 1]
        files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if r['Counter_Name'] != counter:
                continue
            n = r['Kernel_Name']
            if 'msda_fused_fwd' in n or 'msda_win_schedule' in n:
                acc[(key, 'fwd')][counter] += float(r['Counter_Value']) / iters
            elif any(t in n for t in ('msda_fused_bwd', 'msda_tile', 'msda_bin', 'msda_plan', 'msda_grad_finish')):
                acc[(key, 'bwd')][counter] += float(r['Counter_Value']) / iters
    out = {'_note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) of '
                    'tools/prof_msda_single.py <cfg> %d <noise> pair (fused MSDA core as the module calls it: bf16 value / out, fp32 [offsets | logits] rows, '
                    'bf16 gradient rows), MI355X, round-3 kernels (device-planned persistent tile pass).  n0 = offsets of a freshly initialised model (what bench.py runs), '
                    'n1 = ring bias + N(0,1) px ("adapter" offsets).  '
                    'Counter unit KiB; per call = sum over the kernels of one forward / backward call. '
                    'hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction of MI355X_MICROARCH.md).' % iters}
    for (key, d), c in sorted(acc.items()):
        f, w = c.get('FETCH_SIZE', 0.0), c.get('WRITE_SIZE', 0.0)
        out['%s_%s' % (key, d)] = {'FETCH_SIZE_KiB': round(f, 1), 'WRITE_SIZE_KiB': round(w, 1),
                                   'hbm_bytes_per_launch': int(2 * f * 1024 + w * 1024)}
    # the keys bench.py reads: the bench-like (n0) numbers
    for cfg in ('cfg3_inj', 'cfg3_ext'):
        for d in ('fwd', 'bwd'):
            k = '%s_n0_%s' % (cfg, d)
            if k in out:
                out['%s_%s' % (cfg, d)] = out[k]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out['msda_source_digest'] = bench.msda_source_digest()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
