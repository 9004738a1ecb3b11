"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) into a per-kernel text table.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), '
                       'max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) '
                       'from kernels group by name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('# source: %s (rocprofv3 --kernel-trace --stats)' % path)
    print('%-100s %7s %12s %10s %10s %10s %6s %5s %5s %7s %8s %5s' % (
        'kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'agpr', 'lds', 'grid_x', 'wg_x'))
    for r in rows:
        print('%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %7d %8d %5d' % (
            r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))


if __name__ == '__main__':
    main(sys.argv[1])
