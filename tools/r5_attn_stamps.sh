#!/bin/bash
# lab build of the persistent attention kernel with phase stamps + the stamp dump.   usage (GPU box): bash tools/r5_attn_stamps.sh
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python tools/attn_stamps.py 196
