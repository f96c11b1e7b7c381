#!/usr/bin/env python
"""Average per-dispatch counter values per kernel from rocprofv3 --pmc csv output: pmc_summary.py <dir> [substring]"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*$', '', re.sub(r'\(anonymous namespace\)::|vah::|void ', '', r['Kernel_Name']))[:70]
        if pat not in name:
            continue
        a = acc[name][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        s, n = acc[name][c]
        print('   %-30s %16.0f  (avg of %d dispatches)' % (c, s / n, n))
