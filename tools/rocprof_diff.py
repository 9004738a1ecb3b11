"""Per-kernel difference of two rocprofv3 --kernel-trace results (rocpd .db) of the same workload run two ways: calls and total time per kernel
name in each, sorted by the time difference.   python tools/rocprof_diff.py <a.db> <b.db> [steps_a steps_b]"""
import sqlite3
import sys


def load(path):
    cur = sqlite3.connect(path).cursor()
    return {n: (c, t / 1e3) for n, c, t in cur.execute('select name, count(*), sum(end-start) from kernels group by name')}


a, b = load(sys.argv[1]), load(sys.argv[2])
sa, sb = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (1.0, 1.0)
rows = []
for n in set(a) | set(b):
    ca, ta = a.get(n, (0, 0.0))
    cb, tb = b.get(n, (0, 0.0))
    rows.append((tb / sb - ta / sa, n, ca / sa, ta / sa, cb / sb, tb / sb))
rows.sort(reverse=True)
print('%-90s %9s %10s %9s %10s %10s' % ('kernel (per step)', 'calls A', 'us A', 'calls B', 'us B', 'B - A us'))
for d, n, ca, ta, cb, tb in rows[:25] + rows[-8:]:
    print('%-90s %9.1f %10.1f %9.1f %10.1f %+10.1f' % (n[:90], ca, ta, cb, tb, d))
print('total per step: A %.1f us, B %.1f us' % (sum(v[1] for v in a.values()) / sa, sum(v[1] for v in b.values()) / sb))
