"""Summary of DL_STK_STAMPS output (dl_emulated_stacked_kernel): median over workgroups of the s_memtime difference between consecutive stamps, in units of 100 shader-clock
cycles (the shader clock runs at ~2.2 GHz under this load: 100 units = 4.5 us)."""
import sys
import numpy as np

rows, launches = [], []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        if rows: launches.append(np.array(rows, dtype='f8')); rows = []
        continue
    rows.append([int(v) for v in line.split()])
for il, a in enumerate(launches):
    names = ['entry', 'inputs', 'monomials'] + [n for gi in range(13) for n in ('g%d networks' % gi, 'g%d gemm' % gi)] + ['', 'stored', 'realtime']
    live = [q for q in range(15) if np.all(a[:, q] > 0)] + [30]
    if np.all(a[:, 15:30] > 0):
        d = np.median(np.diff(a[:, 15:30], axis=1), axis=0)
        print('   first group, layers 1-3 (cycles): ' + ' | '.join('MFMA %d, barrier %d, act %d, barrier %d' % tuple(d[5 * l:5 * l + 4]) + (', next %d' % d[5 * l + 4] if 5 * l + 4 < len(d) else '') for l in range(3)))
    t0 = a[:, live[0]]
    print('launch %d: %d workgroups; per-workgroup medians (units of 100 shader cycles):' % (il, len(a)))
    prev = live[0]
    for q in live[1:]:
        print('   %-16s +%8.2f   (at %8.2f)' % (names[q], np.median(a[:, q] - a[:, prev]) / 100., np.median(a[:, q] - t0) / 100.))
        prev = q
    print('   (values: shader-clock cycles / 100; s_memtime counters differ between XCDs: only differences within a workgroup mean something)')
