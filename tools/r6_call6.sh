#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --no-cpu --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']; print('headline ms', round(d['ms_per_step'], 3), 'frac', round(r['frac'], 4), 'attainable', r.get('attainable'), 'foa', r.get('frac_of_attainable'), 'sclk', r.get('sclk_mhz_observed')); print(json.dumps(r.get('other_launches')), r.get('step_coverage'))"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for mode in ddp reducer; do
  if [ $mode = ddp ]; then EXTRA="--wrap ddp"; else EXTRA="--batchnorm sync --always-bucket"; fi
  python bench.py --workload whmr_train $EXTRA --no-cpu --steps 10 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$mode single-process rccl1 ms', round(d['ms_per_step'],3))"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/r6_prof_ddp -o ddp -- python3 $R/bench.py --workload whmr_train --wrap ddp --no-cpu --steps 4 --warmup 2 > $OUT/r6_prof_ddp.log 2>&1
DB=$(find $OUT/r6_prof_ddp -name '*.db' | head -1)
python3 $R/tools/rocprof_summary.py $DB | cut -c1-190 | head -45 > $OUT/r6_ddp_kernel_stats.txt
rm -rf $OUT/r6_prof_ddp
head -45 $OUT/r6_ddp_kernel_stats.txt
