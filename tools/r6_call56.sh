#!/bin/bash
# round 6, call 56: shared side-stream pool -- whole GPU suite, default bench (secondary legs) vs dedicated runs
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py 2>/dev/null | python -c "import sys,json; [print('default bench: headline', round(json.loads(l)['ms_per_step'],3), 'x3', round(json.loads(l)['secondary']['vit224_bf16x3']['ms_per_step'],3), 'whmr', round(json.loads(l)['secondary']['whmr']['ms_per_step'],3), 'whmr_train', round(json.loads(l)['secondary']['whmr_train']['ms_per_step'],3), 'sclk', json.loads(l)['roofline'].get('sclk_mhz_observed')) for l in sys.stdin if l.startswith('{')]"
python bench.py --workload whmr_train --no-cpu --no-ceilings --steps 30 --warmup 30 2>/dev/null | python -c "import sys,json; [print('dedicated whmr_train', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
python bench.py --workload whmr --no-cpu --no-ceilings 2>/dev/null | python -c "import sys,json; [print('dedicated whmr', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
