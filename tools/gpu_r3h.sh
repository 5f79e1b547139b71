#!/bin/bash
# round-2 GPU call AH: seeded random small configurations and the late reference tests through HIP
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_fuzz_small_configurations.py tests/test_reference_models.py tests/test_reference_grids.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log | cut -c1-300
