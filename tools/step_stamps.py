"""Summary of DL_STEP_STAMPS output (dl_step_kernel): per workgroup s_memtime at 0 entry, 1 theory done, 2 published, 3 rows of the row block ready, 4 GEMM + finalize done;
6 / 7: s_memrealtime (100 MHz) at entry / exit.  Medians over the workgroups, in microseconds of the measured shader clock."""
import sys
import numpy as np

rows, launches = [], []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        if rows: launches.append(np.array(rows, dtype='f8')); rows = []
        continue
    rows.append([int(v) for v in line.split()])
for il, a in enumerate(launches):
    real = (a[:, 7] - a[:, 6]) / 100.          # us per workgroup
    clk = np.median((a[:, 4] - a[:, 0]) / real)   # shader cycles per us
    t0 = a[:, 0].min()
    print('launch %d: %d workgroups, shader clock %.0f MHz; span of the launch (first entry -> last exit, 100 MHz clock): %.2f us' % (il, len(a), clk, (a[:, 7].max() - a[:, 6].min()) / 100.))
    names = ['entry (after the first)', 'theory done', 'published', 'rows ready', 'GEMM + finalize done']
    for q in range(5):
        d = (a[:, q] - t0) / clk
        print('   %-24s median %7.2f  min %7.2f  max %7.2f us' % (names[q], np.median(d), d.min(), d.max()))
    print('   phases (median): theory %.2f, publish %.2f, wait %.2f, GEMM + finalize %.2f us' % tuple(np.median((a[:, q + 1] - a[:, q]) / clk) for q in range(4)))
