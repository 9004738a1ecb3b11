#!/bin/bash
# Phase stamps of the persistent attention kernel.  `build` (run where hipcc is, e.g. in the build container before gpurun) makes the lab library
# tools/lab/libwhmr_hip_att16_stamps.so = the product objects with csrc/attention_blk16.hip recompiled under -DATT16_STAMPS; without an argument
# (GPU box) it dumps the stamps.     usage: bash tools/r5_attn_stamps.sh build ; gpurun -- bash tools/r5_attn_stamps.sh
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
if [ "${1:-}" = "build" ]; then
  python -m whmr_amd.build > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DATT16_STAMPS -c w-hmr_amd/csrc/attention_blk16.hip -o /tmp/attention_blk16_stamps.o
  OBJS=$(ls w-hmr_amd/build/*.o | grep -v "/attention_blk16.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libwhmr_hip_att16_stamps.so $OBJS /tmp/attention_blk16_stamps.o
  ls -la tools/lab/libwhmr_hip_att16_stamps.so
  exit 0
fi
python tools/attn_stamps.py 196
