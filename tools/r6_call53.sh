#!/bin/bash
# round 6, call 53: co-residency probe (evidence), determinism 3x, whole GPU suite, smoke
mkdir -p gpurun_out
python tools/r6_coresidency_probe.py 10 2>&1 | grep -v Warning | grep -v amdgpu.ids > gpurun_out/r6_coresidency_probe.txt
cat gpurun_out/r6_coresidency_probe.txt | cut -c1-200
for i in 1 2 3; do python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "batch64_bf16_finite" 2>&1 | tail -1; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
