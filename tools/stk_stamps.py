"""Summary of DL_STK_STAMPS output (dl_emulated_stacked_kernel): median over workgroups of the s_memtime difference between consecutive stamps, in units of 100 shader-clock
cycles (the shader clock runs at ~2.2 GHz under this load: 100 units = 4.5 us)."""
import sys
import numpy as np

rows, launches = [], []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        if rows: launches.append(np.array(rows, dtype='u8' if len(rows[0]) == 128 else 'f8')); rows = []
        continue
    rows.append([int(v) for v in line.split()])
for il, a in enumerate(launches if not (launches and launches[0].shape[1] == 128) else []):
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


def overlapped(launches):
    """dl_emulated_stacked_ov_kernel (round 6): 128 slots per workgroup -- 64 of wave 0 then 64 of wave 4; slots 0-2 raw s_memtime (entry / inputs / monomial rows, wave 0),
    then (code << 56 | s_memtime) in program order of each wave; slot 63: HW_ID."""
    codes = {0x1: 'batch in place', 0x2: 'slot 1 work done', 0x3: 'slot 1 barrier', 0x4: 'slot 2 work done', 0x5: 'slot 2 barrier', 0x6: 'gemm (all waves)', 0x7: 'tail'}
    mask = (1 << 56) - 1

    def label(code):
        if code >= 0x80: return '   layer %d %s' % ((code - 0x80) >> 2, ['mfma done', 'sync', 'activations', 'sync'][code & 3])
        return 'tail done' if code == 0x70 else '%s g%d' % (codes.get(code >> 4, hex(code)), code & 0xf)

    for il, a in enumerate(launches):
        a = a.astype('u8')
        t0 = a[:, 0].astype('f8')
        print('launch %d: %d workgroups; medians over workgroups, units of 100 shader cycles since the entry of wave 0' % (il, len(a)))
        print('   %-28s %9.2f / %9.2f' % ('inputs / monomial rows', np.median(a[:, 1] - t0) / 100., np.median(a[:, 2] - t0) / 100.))
        for w, base in ((0, 0), (4, 64)):
            print('  wave %d:' % w)
            prev = None
            for q in range(3, 63):
                v = a[:, base + q]
                if not np.all(v > 0): break
                d = np.median((v & mask).astype('f8') - t0) / 100.
                print('   %-28s %9.2f   (+%8.2f)' % (label(int(v[0] >> 56)), d, d - prev if prev is not None else 0.))
                prev = d
        simd = lambda h: (int(h) >> 4) & 3
        print('   SIMD of wave 0 / wave 4 (HW_ID bits 5:4), first 8 workgroups: ' + ' '.join('%d/%d' % (simd(a[w, 63]), simd(a[w, 127])) for w in range(min(8, len(a)))))


def split_form(launches):
    """dl_emulated_stacked_gemm_kernel: slots 0-31 as dl_emulated_stacked_kernel (wave 0), slots 32 + 12 w + gi / + 6 + gi: wave w at the start / end of the feature GEMM of group gi."""
    for il, a in enumerate(launches):
        a = a.astype('f8')
        names = ['entry', 'inputs', 'monomials'] + [n for gi in range(13) for n in ('g%d record in place' % gi, 'g%d gemm' % gi)] + ['', 'tail done', 'realtime']
        live = [q for q in range(15) if np.all(a[:, q] > 0)] + [30]
        print('launch %d: %d workgroups; per-workgroup medians (units of 100 shader cycles):' % (il, len(a)))
        prev = live[0]
        for q in live[1:]:
            print('   %-22s +%8.2f   (at %8.2f)' % (names[q], np.median(a[:, q] - a[:, prev]) / 100., np.median(a[:, q] - a[:, live[0]]) / 100.))
            prev = q
        if np.all(a[:, 22:24] > 0):   # inside the entry (early theta rows): descriptors of the inputs loaded, theta rows in LDS
            print('   entry: ' + '  '.join('%s %.1f' % (n, v) for n, v in zip(['-> input descriptors', '-> theta rows in LDS (barrier)'], np.median(a[:, 22:24] - a[:, [0]], axis=0) / 100.)))
        if np.all(a[:, 24:27] > 0):   # wave 7 beside the monomial rows: start, log-priors done, first two records requested
            print('   wave 7 beside the monomial rows: ' + '  '.join('%s %.1f' % (n, v) for n, v in zip(['start', '-> log-priors', '-> records requested'], np.median(a[:, 24:27] - a[:, [0]], axis=0) / 100.)))
        w = a[:, 32:128].reshape(len(a), 8, 12)
        for gi in range(6):
            if not np.all(w[:, :, gi] > 0): continue
            start, end = w[:, :, gi] - a[:, [0]], w[:, :, 6 + gi] - a[:, [0]]
            print('   g%d gemm per wave (column block): start %s | duration %s | end spread (max - min) %.1f' % (
                gi, ' '.join('%.0f' % v for v in np.median(start, axis=0) / 100.), ' '.join('%.0f' % v for v in np.median(end - start, axis=0) / 100.), np.median(end.max(axis=1) - end.min(axis=1)) / 100.))


if __name__ == '__main__' and launches and launches[0].shape[1] == 128:
    (overlapped if np.all(launches[0].astype('u8')[:, 3] >> 56) else split_form)(launches[:2])
