#!/bin/bash
# Copy the summaries tools/gpu_round.sh left under gpurun_out/<tag>_* into profiles/r02_* (headers name the command and the commit).
# usage: bash tools/collect_profiles.sh <tag>
T=${1:-r02f}
H=$(git rev-parse --short HEAD)
cd gpurun_out
hdr() { echo "# $1"; echo "# tree: commit $H; one MI355X gpurun box, $(date -u +%Y-%m-%d); produced by tools/gpu_round.sh $T"; }
{ hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --steps 10 --warmup 3   (ViT-B/16 224^2 batch 64 bf16; 13 steps + 1 instrumented step in the trace)"; tail -n +2 ${T}_vit224_kernel_stats.txt; } > ../profiles/r02_vit224_b64_kernel_stats.txt
{ hdr "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py --no-cpu --steps 3 --warmup 2  (mean per dispatch, summed over the device; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMA count)"; grep -E "gemm_blk|attention|layernorm|patch" ${T}_vit224_pmc_SQ.txt; echo "# --pmc FETCH_SIZE (own pass, KiB, raw)"; grep -E "gemm_blk|attention|layernorm|patch" ${T}_vit224_pmc_FETCH_SIZE.txt; echo "# --pmc WRITE_SIZE (own pass, KiB)"; grep -E "gemm_blk|attention|layernorm|patch" ${T}_vit224_pmc_WRITE_SIZE.txt; } > ../profiles/r02_vit224_gemm_pmc.txt
cp ${T}_vit224_gemm_traffic.json ../profiles/r02_vit224_gemm_traffic.json
{ hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload whmr --no-cpu --no-parity --steps 10 --warmup 3   (full W-HMR forward, batch 64 + one 600x800 frame, bf16, HIP-graph replays + one eager instrumented step + the warm-ups)"; tail -n +2 ${T}_whmr_kernel_stats.txt; } > ../profiles/r02_whmr_b64_kernel_stats.txt
{ hdr "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --workload whmr --eager --no-cpu --no-parity --steps 3 --warmup 2; mean per dispatch, summed over the device, KiB; HBM-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 on gfx950 (MI355X_MICROARCH.md)"; echo "# FETCH_SIZE"; grep -E "maf_sample|smpl_|regressor_|tz_|attention|layernorm_blk" ${T}_whmr_pmc_FETCH_SIZE.txt; echo "# WRITE_SIZE"; grep -E "maf_sample|smpl_|regressor_|tz_|attention|layernorm_blk" ${T}_whmr_pmc_WRITE_SIZE.txt; } > ../profiles/r02_whmr_pmc.txt
{ hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload whmr_train --no-cpu --steps 4 --warmup 2   (W-HMR training step, batch 64, bf16, Adam inside the step)"; tail -n +2 ${T}_whmr_train_kernel_stats.txt; } > ../profiles/r02_whmr_train_b64_kernel_stats.txt
for w in vit224 whmr whmr_train vit256x192 vitl256x192_b32; do cp ${T}_bench_$w.json ../profiles/r02_bench_$w.json; done
cd ..
python - <<'PY'
import json
for f in ['vit224','whmr','whmr_train','vit256x192','vitl256x192_b32']:
    d=json.load(open('profiles/r02_bench_%s.json'%f)); print(f, round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],3), d['roofline'].get('traffic'))
d=json.load(open('profiles/r02_bench_whmr.json')); print(d['parity'], d['fp32_ms_per_step'], {k:(round(v['avg_us'],1), round(v['achieved_GBps'])) for k,v in d['hbm_rows'].items()})
PY
