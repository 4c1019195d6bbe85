"""Idle time per step from a rocprofv3 kernel trace of bench.py: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [n gaps]
A step = from one curvature kernel's start to the next one's; covered = union of the kernels' intervals on all queues."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
starts = [i for i, e in enumerate(ev) if 'curvature_valid' in e[2]]
short = lambda n: re.sub(r'\(.*', '', re.sub(r'loamx::\(anonymous namespace\)::', '', n)).replace('void ', '')[:48]
for a, b in zip(starts[1:-1], starts[2:]):
    seg, t0, t1 = ev[a:b], ev[a][0], ev[b][0]
    cov, cs, ce, gaps = 0, seg[0][0], seg[0][1], []
    for s, e, n in seg[1:]:
        if s > ce:
            gaps.append((s - ce, n)); cov += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    cov += ce - cs
    gaps.append((t1 - ce, 'next step'))
    print('step %.3f ms, covered %.3f ms, idle %.1f us: ' % ((t1 - t0) / 1e6, cov / 1e6, (t1 - t0 - cov) / 1e3) +
          ', '.join('%.1f %s' % (g / 1e3, short(n)) for g, n in sorted(gaps, reverse=True)[:top]))
