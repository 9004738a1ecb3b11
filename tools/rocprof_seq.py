"""Print the last N kernel launches of a rocprofv3 --kernel-trace result (rocpd .db) in start order, with gaps.

    python tools/rocprof_seq.py gpurun_out/prof/x_results.db 60
"""
import sqlite3
import sys


def main(path, n):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size from kernels order by start desc limit %d' % n).fetchall()[::-1]
    prev_end = None
    tot = 0
    for name, st, en, gx, gy, gz, wx, lds in rows:
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        prev_end = en
        tot += en - st
        short = name.split('(')[0][-70:]
        print('%8.1f us  gap %6.1f  grid %7d x%d x%d wg %4d lds %6d  %s' % ((en - st) / 1e3, gap, gx // max(wx, 1), gy, gz, wx, lds, short))
    print('busy %.1f us, span %.1f us' % (tot / 1e3, (rows[-1][2] - rows[0][1]) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
