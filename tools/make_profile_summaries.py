"""Turn one `rocprofv3 --kernel-trace --stats -d DIR -- python bench.py` run into the committed summaries:
profiles/r01_bench_final_kernel_stats.csv (first 120 rows of the stats), r01_bench_final_msda_rows.txt (MSDA
kernel averages vs bench.py's event-timed roofline kernel) and r01_bench_last_step_by_kernel.txt (steady-state
step: per-kernel table + kernel families).

    python tools/make_profile_summaries.py gpurun_out/prof_final "<note about the bench lines of the run>"
"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(n):
    if n.startswith('Cijk') or n.startswith('Custom_Cijk'):
        return 'hipblaslt gemm ' + ('wgrad(BSS)' if '_BSS_' in n else 'fwd/dgrad(BBS)')
    if 'reduce_splits' in n:
        return 'gemm split-K reduce'
    if 'attn' in n:
        return 'flash attention (ours)'
    if 'msda' in n:
        return 'msda (ours)'
    if 'ln_' in n or 'scale_' in n or 'finalize_partials' in n or 'colsum' in n or 'dwconv' in n:
        return 'row-streaming kernels (ours)'
    if 'tail_' in n or 'bn_finalize' in n or 'transpose_tokens' in n or 'maxpool' in n:
        return 'batchnorm tail / bn+relu / layout / pool (ours)'
    if 'multi_tensor' in n:
        return 'optimizer'
    if 'BatchNorm' in n:
        return 'miopen batchnorm'
    if 'conv' in n.lower() or 'igemm' in n or 'Im2d' in n or 'Col2Im' in n or 'transpose' in n or 'SubTensor' in n:
        return 'miopen/ck conv'
    if n.startswith('void at::') or n.startswith('at::'):
        return 'torch elementwise/reduce/copy'
    return 'other'


def main():
    d, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    stats = glob.glob(os.path.join(d, '*', '*kernel_stats.csv'))[0]
    trace = glob.glob(os.path.join(d, '*', '*kernel_trace.csv'))[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(ROOT, 'profiles', 'r01_bench_final_kernel_stats.csv'), 'w', newline='') as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows[:120]:
            w.writerow(r)
    ms = [r for r in rows if 'msda' in r['Name']]

    def avg(pat, lv):
        for r in ms:
            if pat in r['Name'] and ('Li%dELi4' % lv in r['Name'] or (lv == 1 and 'int, E, 4' in r['Name'])):
                return float(r['AverageNs']) / 1e3
        return float('nan')
    a = [avg('bwd_vec4', 1), avg('bwd_vec4', 3), avg('gv_mfma', 1), avg('gv_mfma', 3)]
    txt = ['rocprofv3 --kernel-trace --stats -- python bench.py   (default arguments: 3 warm-up + 10 timed steps, MI355X)',
           note,
           'roofline.kernel msda_fused_bwd (HIP events around one vah_msda_fused_backward call) = the two kernels below +',
           'the grad_value handling, 6 extractor (L=1) : 4 injector (L=3) calls per step.', '',
           '%-100s %8s %10s' % ('kernel', 'calls', 'avg us')]
    for r in ms:
        txt.append('%-100s %8s %10.1f' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3))
    txt += ['', 'per backward call: vec4 %.1f (ext) / %.1f (inj) + gv_mfma %.1f (ext) / %.1f (inj)  ->  '
            '(6*(%.1f) + 4*(%.1f)) / 10 = %.1f us' % (a[0], a[1], a[2], a[3], a[0] + a[2], a[1] + a[3],
                                                       (6 * (a[0] + a[2]) + 4 * (a[1] + a[3])) / 10)]
    open(os.path.join(ROOT, 'profiles', 'r01_bench_final_msda_rows.txt'), 'w').write('\n'.join(txt) + '\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_last_step.py'), trace, '--top', '70'],
                         capture_output=True, text=True).stdout
    tr = list(csv.DictReader(open(trace)))
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(tr) if 'multi_tensor_apply_kernel' in r['Kernel_Name']]
    groups, prev = [], None
    for i in marks:
        if prev is None or i - prev > 5:
            groups.append([i, i])
        else:
            groups[-1][1] = i
        prev = i
    step = tr[groups[-2][1] + 1:groups[-1][1] + 1]
    fam = defaultdict(lambda: [0, 0.0])
    for r in step:
        f = fam[family(r['Kernel_Name'])]
        f[0] += 1
        f[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    tot = sum(v[1] for v in fam.values())
    lines = ['kernel families of one steady-state step (same trace), %d launches, %.2f ms busy' % (len(step), tot)]
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        lines.append('  %-48s %5d launches %7.2f ms %5.1f%%' % (k, c, t, 100 * t / tot))
    open(os.path.join(ROOT, 'profiles', 'r01_bench_last_step_by_kernel.txt'), 'w').write(out + '\n' + '\n'.join(lines) + '\n')
    print('\n'.join(txt[-2:]))
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
