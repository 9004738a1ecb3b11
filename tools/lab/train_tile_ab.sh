#!/bin/bash
# Which row-major tile should the N = 2304 / N = 3072 launches of the training step take?  (options 100 / 102 of whmr_set_option force a tile id of
# gemm_bf16_big for that N; chooser = 0)  Training step, interleaved on one box.  usage: train_tile_ab.sh "<ids for N=2304>" "<ids for N=3072>"
cd ${GRAFT_REPO_ROOT:-$PWD}
fmt() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms/step', round(d['ms_per_step'],3))"; }
opt() { python tools/lab/run_with_option.py $1 $2 --workload whmr_train --no-cpu --steps 20 --warmup 5 2>/dev/null | fmt "option $1 = $2"; }
A=${1:-"256 258 259 192 194"}; B=${2:-"257 258 256 194 320"}
for i in 1 2; do
  opt 100 0
  for t in $A; do opt 100 $t; done
  for t in $B; do opt 102 $t; done
done
