#!/bin/bash
# Round 5 probe record on ONE box: attention kernels isolated (bf16, bf16x3) and inside the model (interleaved A/B), the attention phase stamps,
# the SMPL call per batch size.     usage (GPU box): bash tools/r5_probes.sh <commit>  ->  gpurun_out/profiles_r05/r05_attention_probes.txt, r05_smpl_timing.txt
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/profiles_r05; mkdir -p $OUT; cd $R
C=${1:-unknown}
hdr() { echo "# $1"; echo "# tree: commit $C; one MI355X gpurun box, $(date -u +%Y-%m-%d); produced by tools/r5_probes.sh"; }
{ hdr "python tools/attn_probe.py; python tools/x3_attn_probe.py; python tools/r5_attn_ab.py; bash tools/r5_attn_stamps.sh"
  echo "## isolated, bf16 (tools/attn_probe.py)"; timeout 300 python tools/attn_probe.py 2>&1 | grep -v "^/opt\|Warning\|warn"
  echo "## isolated, bf16x3 (tools/x3_attn_probe.py)"; timeout 300 python tools/x3_attn_probe.py 2>&1 | grep -v "^/opt\|Warning\|warn"
  echo "## inside the model, three interleaved rounds per workload (tools/r5_attn_ab.py; 'old' = WHMR_ATTN_OLD kernels)"; timeout 600 python tools/r5_attn_ab.py 2>&1 | grep -v "^/opt\|Warning\|warn"
  echo "## phase stamps of the persistent bf16 kernel (lab build -DATT16_STAMPS, tools/attn_stamps.py 196)"; timeout 600 bash tools/r5_attn_stamps.sh 2>&1 | grep -v "^/opt\|Warning\|warn" | head -48
} > $OUT/r05_attention_probes.txt
{ hdr "python tools/smpl_timing.py"; timeout 300 python tools/smpl_timing.py 2>&1 | grep -v "^/opt\|Warning\|warn"; } > $OUT/r05_smpl_timing.txt
wc -l $OUT/r05_attention_probes.txt $OUT/r05_smpl_timing.txt
