#!/bin/bash
# A/B of the TN weight-gradient grid mapping (XCD-aware vs dispatch order), isolated and inside the training step
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/tn_xcd_probe.py 2>/dev/null
WHMR_TN_RR=1 python tools/tn_xcd_probe.py 2>/dev/null
run() { python bench.py --workload whmr_train --no-cpu --no-secondary --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 whmr_train ms', round(d['ms_per_step'],3))"; }
run xcd
WHMR_TN_RR=1 run rr
run xcd
WHMR_TN_RR=1 run rr
