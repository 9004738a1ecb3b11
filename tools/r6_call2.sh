#!/bin/bash
# Round 6, second GPU call: split-K composed Tz conv in the model, bench line with ceilings / clock, full-forward timeline under rocprofv3.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "tz_" 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 > $OUT/r6_bench_vit224.json 2> $OUT/r6_bench_vit224.err; tail -2 $OUT/r6_bench_vit224.err
python - <<PY
import json
d = json.load(open('$OUT/r6_bench_vit224.json'))
print('ms', d['ms_per_step'], 'roofline', {k: v for k, v in d['roofline'].items() if k not in ('kernel', 'traffic_note', 'attainable_note')})
for k, v in d.get('secondary', {}).items():
    if isinstance(v, dict): print(k, v.get('ms_per_step'), v.get('sclk_mhz_observed'), v.get('hbm_rows'))
print(d['cpu_baseline']['cpu'])
PY
for i in 1 2; do
  python bench.py --workload whmr --no-cpu --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whmr bf16 ms', d['ms_per_step'], d['roofline'].get('sclk_mhz_observed'))"
done
python bench.py --workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whmr x3 ms', d['ms_per_step'], d['roofline'].get('sclk_mhz_observed'))"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/r6_prof_whmr -o whmr -- python3 $R/bench.py --workload whmr --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/r6_prof_whmr.log 2>&1
DB=$(find $OUT/r6_prof_whmr -name '*.db' | head -1)
python3 $R/tools/whmr_timeline.py $DB 0 > $OUT/r6_whmr_timeline.txt 2>&1
rm -rf $OUT/r6_prof_whmr
head -5 $OUT/r6_whmr_timeline.txt; tail -6 $OUT/r6_whmr_timeline.txt
