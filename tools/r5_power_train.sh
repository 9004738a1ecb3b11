#!/bin/bash
# Package clock / power under the other bench loops (training step with grouped / single weight-gradient launches, full forward in both numerics):
# does any of them sit at a depressed clock below the power cap like the phase-locked attention kernel did?   usage: bash tools/r5_power_train.sh
. $(dirname $0)/power_lib.sh
python -c "import torch" 2>/dev/null
smp() {  # label, env, args   (bench line without --no-secondary: these workloads have none)
  echo "## $1"
  env $2 python bench.py --no-cpu $3 --warmup 5 > /tmp/b.json 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz)/sclk \1 MHz/' -e 's/.*Power (W): \([0-9.]*\)/  power \1 W/' | tr '\n' ' ' | awk '$5+0 > 400 || $2+0 > 600 {print}'
    sleep 0.3
  done
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json
}
smp "whmr_train (default)" "WHMR_X=0" "--workload whmr_train --steps 500"
smp "whmr_train, single weight-gradient launches" "WHMR_TN_GROUP=0" "--workload whmr_train --steps 500"
smp "whmr forward bf16 (graph replay)" "WHMR_X=0" "--workload whmr --no-parity --steps 2500"
smp "whmr forward bf16x3 (graph replay)" "WHMR_X=0" "--workload whmr --numerics bf16x3 --no-parity --steps 1000"
smp "vit224 fp32 parity mode" "WHMR_X=0" "--numerics fp32 --no-secondary --steps 400"
