#!/bin/bash
# kernel trace of the training step -> phase / stream timeline (tools/train_timeline.py)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python3 $R/bench.py --workload whmr_train --no-cpu --no-secondary --steps 4 --warmup 2 > /tmp/tl.log 2>&1
DB=$(find /tmp/tl -name '*.db' | head -1)
python3 $R/tools/train_timeline.py $DB
python3 $R/tools/train_timeline.py $DB names | sort -k2 -n -r | head -60
