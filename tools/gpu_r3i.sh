#!/bin/bash
# round-2 GPU call AI: medium-shape random configurations through HIP
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_fuzz_small_configurations.py -m gpu -q -k medium > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log | cut -c1-400
