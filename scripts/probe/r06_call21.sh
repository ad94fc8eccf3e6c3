#!/bin/bash
# round 6, call 21: the whole GPU suite + smoke on the final tree
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06_c21_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r06_c21_tests.txt
