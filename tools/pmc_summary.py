"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).

Per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B,
so the read side is doubled (upper bound for narrower accesses); WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import sys
from collections import defaultdict


def read(dirname, counter):
    sums, counts = defaultdict(float), defaultdict(int)
    for fn in glob.glob(dirname + '/**/*counter_collection.csv', recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row.get('Counter_Name') != counter: continue
                name = row['Kernel_Name'].split('(')[0]
                sums[name] += float(row['Counter_Value'])
                counts[name] += 1
    return {name: sums[name] / counts[name] for name in sums}, counts


fetch, nf = read(sys.argv[1], 'FETCH_SIZE')
write, nw = read(sys.argv[2], 'WRITE_SIZE')
print('%-60s %10s %14s %14s %16s' % ('kernel', 'launches', 'FETCH_SIZE KiB', 'WRITE_SIZE KiB', 'HBM bytes/launch'))
for name in sorted(set(fetch) | set(write)):
    if not name.startswith(('dl_', 'void dl_')): continue
    f, w = fetch.get(name, 0.), write.get(name, 0.)
    print('%-60s %10d %14.1f %14.1f %16.0f' % (name[:60], nf.get(name, 0), f, w, (2. * f + w) * 1024.))
print('HBM bytes/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction of the read side, see the module docstring)')
