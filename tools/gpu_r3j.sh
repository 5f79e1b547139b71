#!/bin/bash
# round-2 GPU call AJ: longer random sweep through HIP (medium shapes, seeds 40..339)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
timeout -k 10 1000 python tools/fuzz_gpu.py 40 300 medium > $O/fuzz_medium.log 2>&1; echo "rc=$?"
tail -12 $O/fuzz_medium.log | cut -c1-700
