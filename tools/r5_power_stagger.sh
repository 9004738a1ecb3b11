#!/bin/bash
# Does starting the CUs of the persistent split-bf16 attention kernel apart (quarters of each XCD, 0.5 us * k) keep the package clock up?
# (The bf16 kernel and the blocked GEMMs were swept the same way with lab patches: no effect -- profiles/r05_power_clock_attention_stagger.txt,
# r05_power_gemm_stagger_lab.txt.)      usage (GPU box): bash tools/r5_power_stagger.sh > gpurun_out/r05_power_stagger.txt
. $(dirname $0)/power_lib.sh
python -c "import torch" 2>/dev/null
X="--numerics bf16x3 --steps 1500"
for w in vit224 vit256x192; do
smp "bf16x3 $w, old attention" "WHMR_ATTN_OLD=1" "$X --workload $w"
for k in 0 1 2 3 4; do
smp "bf16x3 $w, new attention, stagger $k" "WHMR_ATTN_X3_STAGGER=$k" "$X --workload $w"
done
done
