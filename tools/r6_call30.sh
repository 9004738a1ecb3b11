#!/bin/bash
# round 6, call 30: MapForkFn -- whole GPU suite + smoke
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "map_fork" 2>&1 | tail -3
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
