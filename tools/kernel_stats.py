"""Short table of a rocprofv3 ``*_kernel_stats.csv``: kernel (template arguments kept, argument list dropped), calls, average / total microseconds.
    python tools/kernel_stats.py <file-or-dir>"""
import csv, glob, os, sys
paths = []
for arg in sys.argv[1:]:
    paths += glob.glob(os.path.join(arg, '**', '*kernel_stats.csv'), recursive=True) if os.path.isdir(arg) else glob.glob(arg)
for path in paths:
    for row in csv.DictReader(open(path)):
        name = row['Name'].replace('void ', '')
        depth, cut = 0, len(name)
        for i, ch in enumerate(name):
            if ch == '<': depth += 1
            elif ch == '>': depth -= 1
            elif ch == '(' and depth == 0: cut = i; break
        print('%-72s n = %6d  avg %9.2f us  total %10.1f us  %5.1f %%' % (name[:cut][:72], int(row['Calls']), float(row['AverageNs']) / 1e3, float(row['TotalDurationNs']) / 1e3, float(row['Percentage'])))
