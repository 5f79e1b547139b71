#!/bin/bash
# round-2 GPU call N: the step-graph tests (call M forgot -m gpu) + the parity file with graphs active
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2n
mkdir -p $O
cd $R
OCNHIP_DEBUG=1 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_graph.log 2>&1; echo "pytest rc=$?" >> $O/pytest_graph.log
tail -5 $O/pytest_graph.log
grep -c "step graphs off" $O/pytest_graph.log
