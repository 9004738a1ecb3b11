#!/bin/bash
# round 6, call 35: whole GPU suite + smoke with the Tz tail stream and the heavy-first order (off under capture)
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
