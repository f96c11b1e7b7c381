#!/usr/bin/env python
"""Print the per-kernel rows of a rocprofv3 --stats csv with short names: kstats.py <dir> [substring]"""
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for f in sorted(glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True)):
    print('#', os.path.relpath(f, root))
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::|vah::|void ', '', r['Name'])
        name = re.sub(r'\(.*$', '', name)[:90]
        if pat in name:
            print('%-90s calls %4s avg %9.1f us  min %9.1f  max %9.1f' % (
                name, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
