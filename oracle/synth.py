"""TEST INFRASTRUCTURE ONLY -- re-export of the synthetic weight / asset / input generator.

The generator itself (numpy PCG64 keyed by crc32(name), no compute path) lives in the package (``whmr_amd.utils.synth``) because
``bench.py`` and the timing tools need the same tensors for the GPU leg and nothing but tests / smoke / the CPU-baseline leg may
import from ``oracle/``.  Tests and ``tests/golden/make_golden.py`` keep importing it from here.
"""
from whmr_amd.utils.synth import *                                     # noqa: F401,F403
from whmr_amd.utils.synth import _rng, _t, _normal, _uniform, _linear, _ln, _bn, _resnet50   # noqa: F401
