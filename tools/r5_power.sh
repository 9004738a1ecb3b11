#!/bin/bash
# Power and clock of the package while the bench loops run: is the ViT forward at the power cap, and what do the attention kernels do to the clock?
# usage (GPU box): bash tools/r5_power.sh > gpurun_out/r05_power.txt
. $(dirname $0)/power_lib.sh
python -c "import torch" 2>/dev/null
rocm-smi --showmaxpower 2>/dev/null | grep -i "power (w)"
smp "bf16 vit224 (new attention)" "WHMR_ATTN_OLD=0" "--steps 4000"
smp "bf16 vit224 (old attention)" "WHMR_ATTN_OLD=1" "--steps 4000"
smp "bf16x3 vit224, old attention" "WHMR_ATTN_OLD=1" "--numerics bf16x3 --steps 1800"
smp "bf16x3 vit224, new attention" "WHMR_ATTN_OLD=0" "--numerics bf16x3 --steps 1800"
smp "bf16x3 vit224, old attention" "WHMR_ATTN_OLD=1" "--numerics bf16x3 --steps 1800"
smp "bf16x3 vit224, new attention" "WHMR_ATTN_OLD=0" "--numerics bf16x3 --steps 1800"
