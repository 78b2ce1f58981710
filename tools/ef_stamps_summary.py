"""Summarise DL_EF_STAMPS output (dl_emulated_feature_gram_kernel): per launch, phase durations of the workgroups (shader-clock ticks -> us at 2.4 GHz).
Slots: 0 entry, 1 forward done (records in LDS), 2 main loop of monomial group 0 done, 3 its epilogue done, 4 main loop of group 1 done, 5 its epilogue done (before the
barrier), 6 barrier passed, 7 Gram matrices written."""
import sys
import numpy as np
blocks, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        if cur: blocks.append(np.array(cur, dtype='f8')); cur = []
    else:
        cur.append([float(v) for v in line.split()])
names = ['forward (MLP engines, monomials, records)', 'feature GEMM, monomials 0-9', 'epilogue 0', 'feature GEMM, monomials 10-18', 'epilogue 1', 'barrier', 'Gram matrices']
for ib, a in enumerate(blocks):
    fwd = a[:, 8:14] if a.shape[1] > 8 else None
    if a.shape[1] >= 16 and (a[:, 15] > a[:, 14]).all():   # s_memrealtime (100 MHz) at entry / exit against s_memtime: the shader clock during the kernel
        ghz = np.median((a[:, 7] - a[:, 0]) / ((a[:, 15] - a[:, 14]) * 10.))
        span = (a[:, 15].max() - a[:, 14].min()) * 0.01
        print('launch %d: shader clock %.3f GHz (the times below assume 2.4); first entry -> last exit on the chip-wide clock: %.2f us' % (ib, ghz, span))
    a = a[:, :8]
    d = np.diff(a, axis=1) / 2.4e3
    print('launch %d: %d workgroups; life median %.2f us max %.2f us' % (ib, len(a), np.median((a[:, 7] - a[:, 0]) / 2.4e3), ((a[:, 7] - a[:, 0]) / 2.4e3).max()))
    for q, name in enumerate(names):
        print('  %-44s median %6.2f  p90 %6.2f  max %6.2f us' % (name, np.median(d[:, q]), np.percentile(d[:, q], 90), d[:, q].max()))
    if fwd is not None:   # inside the forward pass: entry barrier, then the barrier of every layer
        used = [q for q in range(fwd.shape[1]) if (fwd[:, q] > 0).all()]
        prev = a[:, 0]
        for n, q in enumerate(used):
            print('    forward: %-36s median %6.2f us' % ('entry -> inputs in LDS (barrier)' if n == 0 else 'layer %d done (barrier)' % (n - 1), np.median((fwd[:, q] - prev) / 2.4e3)))
            prev = fwd[:, q]
        if used: print('    forward: %-36s median %6.2f us' % ('last barrier -> records complete', np.median((a[:, 1] - prev) / 2.4e3)))
    if fwd is not None and (fwd > 0).all():   # temporary: the timeline of wave 4 (team 1) relative to the forward's end
        t0 = a[:, 1]
        lab = ['G4 start', 'G4 end', 'E end', 'G8 end', 'E end', 'G7 end']
        print('    wave 4: ' + '  '.join('%s %.2f' % (l, np.median((fwd[:, q] - t0) / 2.4e3)) for q, l in enumerate(lab)))
        print('    wave 0: G8 end %.2f  E end %.2f  G6+E+G5 end %.2f  E end %.2f  barrier passed %.2f' % tuple(np.median((a[:, q] - t0) / 2.4e3) for q in (2, 3, 4, 5, 6)))
