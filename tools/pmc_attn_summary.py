#!/usr/bin/env python
"""Average per-dispatch counter values per kernel from the csv files of tools/pmc_attn.sh."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*$', '', re.sub(r'\(anonymous namespace\)::|vah::|void ', '', r['Kernel_Name']))[:60]
        if 'attn' not in name:
            continue
        a = acc[name][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        s, n = acc[name][c]
        print('   %-30s %16.0f  (avg of %d dispatches)' % (c, s / n, n))
