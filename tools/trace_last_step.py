"""Summarise the LAST optimizer step of a `rocprofv3 --kernel-trace` run of bench.py.

The whole-run kernel_stats.csv mixes MIOpen's find-mode convolutions and TunableOp replays from the
warm-up with the steady state; this script cuts the trace at the fused AdamW launches
(multi_tensor_apply_kernel marks the end of a step) and reports the kernels between the last two.

    python tools/trace_last_step.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv [--top 70]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--top', type=int, default=70)
    ap.add_argument('--marker', default='multi_tensor_apply_kernel')
    ap.add_argument('--gaps', type=int, default=0, help='also list the N largest idle gaps between kernels')
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if args.marker in r['Kernel_Name']]
    # group consecutive marker launches (one optimizer step = several multi-tensor chunks)
    groups, prev = [], None
    for i in marks:
        if prev is None or any(args.marker not in rows[j]['Kernel_Name'] for j in range(prev, i)):
            groups.append([i, i])
        else:
            groups[-1][1] = i
        prev = i
    if len(groups) < 2:
        raise SystemExit('need at least two optimizer steps in the trace')
    lo, hi = groups[-2][1] + 1, groups[-1][1] + 1
    step = rows[lo:hi]
    wall = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
    agg = defaultdict(lambda: [0, 0.0])
    for r in step:
        a = agg[short(r['Kernel_Name'])]
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy = sum(v[1] for v in agg.values()) / 1e3
    print(f'last step: {len(step)} launches, wall {wall:.2f} ms, kernel-busy {busy:.2f} ms')
    print(f'{"kernel":110s} {"calls":>6s} {"avg us":>9s} {"total ms":>9s} {"share":>6s}')
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f'{k:110s} {c:6d} {t / c:9.1f} {t / 1e3:9.3f} {100 * t / 1e3 / busy:5.1f}%')
    if args.gaps:
        gaps = []
        for a, b in zip(step, step[1:]):
            g = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
            if g > 0:
                gaps.append((g, short(a['Kernel_Name'])[:60], short(b['Kernel_Name'])[:60]))
        tot = sum(g for g, _, _ in gaps) / 1e3
        print(f'idle between kernels: {tot:.2f} ms in {len(gaps)} gaps; gaps > 20 us: '
              f'{sum(g for g, _, _ in gaps if g > 20) / 1e3:.2f} ms')
        by_next = defaultdict(lambda: [0, 0.0])
        for g, a, b in gaps:
            by_next[b][0] += 1
            by_next[b][1] += g
        print('idle time by the kernel that follows the gap:')
        for k, (c, t) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:args.gaps]:
            print(f'  {k:60s} {c:5d} gaps {t / 1e3:8.3f} ms  avg {t / c:7.1f} us')


if __name__ == '__main__':
    main()
