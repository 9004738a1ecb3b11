#!/bin/bash
# the three main bench lines -> gpurun_out/<tag>_bench_{vit224,whmr,whmr_train}.json + a one-line digest of each
TAG=${1:-r02g}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_vit224.json 2>/dev/null
python bench.py --workload whmr > gpurun_out/${TAG}_bench_whmr.json 2>/dev/null
python bench.py --workload whmr_train --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_whmr_train.json 2>/dev/null
python - "$TAG" <<'PY'
import json, sys
t = sys.argv[1]
for f in ("vit224", "whmr", "whmr_train"):
    d = json.load(open("gpurun_out/%s_bench_%s.json" % (t, f)))
    rows = d.get("hbm_rows")
    print(f, round(d["ms_per_step"], 3), round(d["value"]), round(d["roofline"]["frac"], 3), d["roofline"].get("traffic"),
          rows and {k: (round(v["avg_us"], 1), round(v["achieved_GBps"])) for k, v in rows.items()})
PY
