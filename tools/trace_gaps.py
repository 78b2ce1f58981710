"""Timeline of consecutive kernel dispatches from a rocprofv3 rocpd database: duration of each kernel and idle gap to the next one."""
import sqlite3, glob, sys
db = glob.glob(sys.argv[1])[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
n0 = len(rows) // 2
prev_end = None
import collections
stats = collections.defaultdict(list)
for name, st, en in rows[n0:n0 + 600]:
    short = name.split('(')[0][:40]
    if prev_end is not None: stats[('gap before', short)].append(st - prev_end)
    stats[('duration', short)].append(en - st)
    prev_end = en
for key in sorted(stats):
    v = stats[key]
    print('%-12s %-42s n=%4d mean %8.0f ns  min %6d  max %6d' % (key[0], key[1], len(v), sum(v) / len(v), min(v), max(v)))
