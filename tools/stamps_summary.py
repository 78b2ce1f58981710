"""Summarise DL_FS_STAMPS output: per launch, distribution of workgroup start times and per-phase durations (shader-clock ticks -> us at 2.4 GHz)."""
import sys
import numpy as np
blocks, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        if cur: blocks.append(np.array(cur, dtype='f8')); cur = []
    else:
        cur.append([float(v) for v in line.split()])
GHZ = 2.4
for ib, a in enumerate(blocks):
    t0 = a[:, 0].min()
    rel = (a - t0) / GHZ / 1e3
    names = ['start', 'phase01', 'convolution', 'coefficients', 'projection', 'store']
    print('launch %d: %d workgroups' % (ib, len(a)))
    print('  workgroup start : min %.2f median %.2f p90 %.2f max %.2f us' % (rel[:, 0].min(), np.median(rel[:, 0]), np.percentile(rel[:, 0], 90), rel[:, 0].max()))
    for q in range(1, 6):
        d = rel[:, q] - rel[:, q - 1]
        print('  %-13s: median %.2f p90 %.2f max %.2f us' % (names[q], np.median(d), np.percentile(d, 90), d.max()))
    print('  workgroup life  : median %.2f max %.2f us; last exit at %.2f us' % (np.median(rel[:, 5] - rel[:, 0]), (rel[:, 5] - rel[:, 0]).max(), rel[:, 5].max()))
    # s_memtime differs between XCDs; slots 6 / 7 hold s_memrealtime (100 MHz, chip-wide) at entry / exit: the dispatch ramp
    if a.shape[1] >= 8:
        st = (a[:, 6] - a[:, 6].min()) * 0.01
        en = (a[:, 7] - a[:, 6].min()) * 0.01
        print('  chip-wide (100 MHz clock): workgroup starts: median %.2f p90 %.2f max %.2f us; exits: median %.2f max %.2f us' % (
            np.median(st), np.percentile(st, 90), st.max(), np.median(en), en.max()))
        print('  start time by workgroup id (every 64th): ' + ' '.join('%.1f' % v for v in st[::64]))
    if a.shape[1] >= 8:
        life = (a[:, 5] - a[:, 0]) / GHZ / 1e3
        q = len(life) // 4
        print('  workgroup life by id quartile: ' + ' '.join('%.2f' % np.median(life[i * q:(i + 1) * q]) for i in range(4)) + ' ; slowest ids: ' + ' '.join(str(i) for i in np.argsort(life)[-12:]))
