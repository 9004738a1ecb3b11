#!/bin/bash
# round 6, call 59: whole GPU suite, then the evidence refresh on cfc4d8b
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
bash tools/gpu_round.sh r06 cfc4d8b
