#!/bin/bash
# SQ counters of the bf16 attention backward (ViT-B training step, batch 64): what bounds it?
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/ab; rocprofv3 --kernel-trace --pmc $C -d /tmp/ab -o ab -- python3 $R/tools/train_timing.py 64 2 > /tmp/ab.log 2>&1
  python3 $R/tools/pmc_summary.py $(find /tmp/ab -name '*.db' | head -1) attention_bwd 2>/dev/null | cut -c1-140
done
