"""Run bench.py with whmr_set_option(key, value) applied first (A/B of tuning switches that have no environment variable):
    python tools/lab/run_with_option.py <key>[,<key>...] <value>[,<value>...] [bench.py arguments ...]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench                      # noqa: E402
from whmr_amd import _lib         # noqa: E402

for k, v in zip(sys.argv[1].split(','), sys.argv[2].split(',')):
    _lib.lib().whmr_set_option(int(k), int(v))
raise SystemExit(bench.main(sys.argv[3:]))
