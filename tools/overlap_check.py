"""From a rocprofv3 --kernel-trace database: how much kernel time runs concurrently (sum of durations vs the union of the busy intervals), per stream."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
key = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
rows = cur.execute('select start, end, name%s from kernels order by start' % ((', ' + key) if key else '')).fetchall()
n = len(rows)
rows = rows[n // 2:]                      # second half of the run: steady state
tot = sum(r[1] - r[0] for r in rows)
busy, cur_s, cur_e = 0, None, None
for r in rows:
    s, e = r[0], r[1]
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print('kernels %d: sum of durations %.2f ms, union of busy intervals %.2f ms, span %.2f ms -> %.1f %% of the kernel time overlapped'
      % (len(rows), tot / 1e6, busy / 1e6, span / 1e6, 100.0 * (tot - busy) / tot))
if key:
    per = {}
    for r in rows:
        per.setdefault(r[3], [0, 0]); per[r[3]][0] += 1; per[r[3]][1] += r[1] - r[0]
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]): print('  %s %s: %d kernels, %.2f ms' % (key, k, c, t / 1e6))
