#!/bin/bash
# Round 5: timing ablations and PMC of the persistent blocked attention kernel (tools/attn_probe.py under rocprofv3).   usage: bash tools/r5_attn_pmc.sh
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python tools/attn_probe.py 2>&1 | grep -v fp32
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/r05_pmc_attn -o pmc -- python3 $R/tools/attn_probe.py > $OUT/r05_pmc_attn.log 2>&1
DB=$(find $OUT/r05_pmc_attn -name '*.db' | head -1)
cd $R
[ -n "$DB" ] && python tools/pmc_summary.py $DB | grep -E "attention_blk16|attention_bf16_chunk_kernel<7, true>|attention_bf16_chunk_kernel<6, true>"
rm -rf $OUT/r05_pmc_attn
