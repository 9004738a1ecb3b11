"""Timeline of ONE full W-HMR forward from a rocprofv3 --kernel-trace database (bench.py --workload whmr): every launch of a middle step with its start
offset, duration and HSA queue (= stream), + busy time per queue.   usage: python tools/whmr_timeline.py <results.db> [min_us]"""
import sqlite3
import sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
key = 'queue_id' if 'queue_id' in cols else 'stream_id'
rows = cur.execute('select start, end, name, %s, grid_x, workgroup_x from kernels order by start' % key).fetchall()
marks = [i for i, r in enumerate(rows) if 'patch_im2col' in r[2]]
mid = len(marks) // 2
lo, hi = marks[mid], marks[mid + 1]
# the camera branch of step k+1 may start before the patch kernel of step k+1: cut at the patch kernel's start time
t0, t1 = rows[lo][0], rows[hi][0]
step = [r for r in rows if r[0] >= t0 - 30e3 and r[0] < t1 - 30e3]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
qs = sorted({r[3] for r in step})
print('step %d of %d: %d launches, %.3f ms between patch kernels; queues %s' % (mid, len(marks), len(step), (t1 - t0) / 1e6, qs))
busy = {}
for s, e, n, q, gx, wx in step:
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
    if (e - s) / 1e3 >= min_us:
        print('%9.1f us  +%7.1f us  q%-3s grid %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, gx // max(wx, 1), n.split('(')[0][-90:]))
for q in qs:
    print('queue %s busy %.1f us' % (q, busy[q]))
