#!/bin/bash
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "tz_composed_convolution_node" 2>&1 | grep -v "^$" | tail -6
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "batch64_bf16_finite" 2>&1 | tail -1; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
