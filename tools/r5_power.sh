#!/bin/bash
# Power and clock of the package while the bench loops run: is the ViT forward at the power cap?
# usage (GPU box): bash tools/r5_power.sh > gpurun_out/r05_power.txt
smp() {  # label, env, args
  echo "## $1"
  env $2 python bench.py --no-cpu --no-secondary $3 --warmup 5 > /tmp/b.json 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz)/sclk \1 MHz/' -e 's/.*Power (W): \([0-9.]*\)/  power \1 W/' | tr '\n' ' ' | awk '$5+0 > 400 || $2+0 > 600 {print}'
    sleep 0.3
  done
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json
}
python -c "import torch" 2>/dev/null
rocm-smi --showmaxpower 2>/dev/null | grep -i "power (w)"
smp "bf16 vit224 (new attention)" "WHMR_ATTN_OLD=0" "--steps 4000"
smp "bf16 vit224 (old attention)" "WHMR_ATTN_OLD=1" "--steps 4000"
smp "bf16x3 vit224, old attention" "WHMR_ATTN_OLD=1" "--numerics bf16x3 --steps 1800"
smp "bf16x3 vit224, new attention" "WHMR_ATTN_OLD=0" "--numerics bf16x3 --steps 1800"
smp "bf16x3 vit224, old attention" "WHMR_ATTN_OLD=1" "--numerics bf16x3 --steps 1800"
smp "bf16x3 vit224, new attention" "WHMR_ATTN_OLD=0" "--numerics bf16x3 --steps 1800"
