#!/bin/bash
python tools/r6_coresidency_probe.py 10 2>&1 | grep "^load on"
for cfg in "1 128" "1 64" "0 128"; do
  set -- $cfg
  echo "== heavy_first=$1 tn_row_pad=$2"
  WHMR_TRAIN_HEAVY_FIRST=$1 WHMR_TN_ROW_PAD=$2 python tools/r6_det_probe.py 9 2>&1 | grep "^run" | sed 's/head keys.*maf_extractor.2.*/STAGE3 DIFFERS/; s/head keys.*//' | sort | uniq -c | cut -c1-80
done
for r in 1 2; do for pad in 64 128; do
  WHMR_TN_ROW_PAD=$pad python bench.py --workload whmr_train --no-cpu --no-ceilings --steps 30 --warmup 30 2>/dev/null | python -c "import sys,json; [print('tn_row_pad=$pad', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done; done
