"""Run bench.py against an EXPERIMENTAL build of the HIP library (timing experiments whose numerics may be garbage; never a product path):
    python tools/lab/run_with_lib.py tools/lab/libwhmr_hip_<tag>.so --no-cpu --no-secondary ..."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench                      # noqa: E402  (puts the package alias in place)
from whmr_amd import _lib         # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
raise SystemExit(bench.main(sys.argv[2:]))
