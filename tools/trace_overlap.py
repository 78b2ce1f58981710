"""Kernel timeline of a rocprofv3 --kernel-trace run (rocpd sqlite or csv): busy time (union of kernel intervals), summed kernel time, per-kernel mean durations, and
how much of the time two or more kernels were running at once.
    python tools/trace_overlap.py <dir-or-glob of *_kernel_trace.csv>"""
import csv, glob, os, sys
import numpy as np

paths = []
for arg in sys.argv[1:]:
    paths += glob.glob(os.path.join(arg, '**', '*kernel_trace.csv'), recursive=True) if os.path.isdir(arg) else glob.glob(arg)
rows = []
for path in paths:
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r.get('Queue_Id', '0'), r.get('Stream_Id', '0')))
rows.sort()
if not rows: sys.exit('no kernel trace rows found in {}'.format(sys.argv[1:]))
# drop the set-up part: keep the last 60 % of the dispatches
rows = rows[int(0.4 * len(rows)):]
start, end = np.array([r[0] for r in rows], dtype='i8'), np.array([r[1] for r in rows], dtype='i8')
events = sorted([(s, 1) for s in start] + [(e, -1) for e in end])
busy = {0: 0, 1: 0, 2: 0}
level, last = 0, events[0][0]
for t, d in events:
    busy[min(level, 2)] += t - last
    level, last = level + d, t
span = end.max() - start.min()
print('dispatches %d over %.3f ms; kernel time summed %.3f ms; idle %.1f %%, one kernel %.1f %%, two or more %.1f %%' % (
    len(rows), span / 1e6, (end - start).sum() / 1e6, 100. * busy[0] / span, 100. * busy[1] / span, 100. * busy[2] / span))
names = sorted(set(r[2] for r in rows))
for name in names:
    d = np.array([r[1] - r[0] for r in rows if r[2] == name])
    print('  %-62s n = %5d  mean %7.2f us  median %7.2f us' % (name, d.size, d.mean() / 1e3, np.median(d) / 1e3))
print('queues:', sorted(set(r[3] for r in rows)), 'streams:', sorted(set(r[4] for r in rows)))
