#!/bin/bash
# round-2 GPU call O: step graphs (fixed expectation), AMD Cb cases, stand-alone fields -- parity + contract files
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2o
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py tests/test_model_contracts.py tests/test_abi.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
