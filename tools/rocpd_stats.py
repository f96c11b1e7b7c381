#!/usr/bin/env python
"""Per-kernel call counts / average durations from a rocprofv3 rocpd (.db) output directory.

    python tools/rocpd_stats.py gpurun_out/<dir> [substring ...]
"""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    pats = sys.argv[2:]
    dbs = glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True)
    assert dbs, 'no *_results.db under %s' % d
    cur = sqlite3.connect(dbs[0]).cursor()
    q = 'select name, count(*), avg(end - start) / 1000.0, sum(end - start) / 1e6 from kernels group by name order by 4 desc'
    print('%-100s %7s %9s %9s' % ('kernel', 'calls', 'avg us', 'total ms'))
    for name, calls, avg, tot in cur.execute(q):
        if pats and not any(p in name for p in pats):
            continue
        print('%-100s %7d %9.1f %9.2f' % (name[:100], calls, avg, tot))


if __name__ == '__main__':
    main()
