"""Run bench.py with whmr_set_option(key, value) applied first (A/B of tuning switches that have no environment variable):
    python tools/lab/run_with_option.py <key> <value> [bench.py arguments ...]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench                      # noqa: E402
from whmr_amd import _lib         # noqa: E402

_lib.lib().whmr_set_option(int(sys.argv[1]), int(sys.argv[2]))
raise SystemExit(bench.main(sys.argv[3:]))
