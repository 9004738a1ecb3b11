"""Summarise rocprofv3 --pmc results (rocpd sqlite) per kernel: mean of each counter over dispatches."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
rows = cur.execute('select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name').fetchall() if 'kernel_name' in cols else []
if not rows:
    print(cols)
for r in rows:
    if len(sys.argv) < 3 or sys.argv[2] in r[0]:
        print('%-70s %-28s %16.1f  (n=%d)' % (r[0][:70], r[1], r[2], r[3]))
