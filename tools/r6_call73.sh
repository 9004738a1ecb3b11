#!/bin/bash
# round 6, call 73: seven VALU files built without packed-fp32 instructions -- same-box A/B against the previous build (tools/lab/build/libwhmr_packed.so), tests
# (tools/lab/build/libwhmr_packed.so = the library built with WHMR_BUILD_PACKED_FP32=1 python -m whmr_amd.build --force, copied aside; lab binaries are not kept in the tree)
j() { python -c "import sys,json; [print('$1', round(json.loads(l)['ms_per_step'],3)) for l in sys.stdin if l.startswith('{')]"; }
for r in 1 2; do
  python bench.py --no-cpu --no-secondary --no-ceilings 2>/dev/null | j "no-packed  vit224"
  python tools/lab/run_with_lib.py tools/lab/build/libwhmr_packed.so --no-cpu --no-secondary --no-ceilings 2>/dev/null | j "packed     vit224"
  python bench.py --workload whmr --no-cpu --no-ceilings --no-parity 2>/dev/null | j "no-packed  whmr"
  python tools/lab/run_with_lib.py tools/lab/build/libwhmr_packed.so --workload whmr --no-cpu --no-ceilings --no-parity 2>/dev/null | j "packed     whmr"
  python bench.py --workload whmr --numerics bf16x3 --no-cpu --no-ceilings --no-parity --steps 10 --warmup 3 2>/dev/null | j "no-packed  whmr x3"
  python tools/lab/run_with_lib.py tools/lab/build/libwhmr_packed.so --workload whmr --numerics bf16x3 --no-cpu --no-ceilings --no-parity --steps 10 --warmup 3 2>/dev/null | j "packed     whmr x3"
  python bench.py --workload whmr_train --no-cpu --no-ceilings --steps 20 --warmup 20 2>/dev/null | j "no-packed  train"
  python tools/lab/run_with_lib.py tools/lab/build/libwhmr_packed.so --workload whmr_train --no-cpu --no-ceilings --steps 20 --warmup 20 2>/dev/null | j "packed     train"
done
