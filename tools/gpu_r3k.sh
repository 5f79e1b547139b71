#!/bin/bash
# round-2 GPU call AK: longer random sweeps through HIP (small shapes seeds 80..879, medium shapes seeds 340..739)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
timeout -k 10 500 python tools/fuzz_gpu.py 80 800 small > $O/fuzz_small.log 2>&1; echo "rc=$?"
tail -6 $O/fuzz_small.log | cut -c1-700
timeout -k 10 500 python tools/fuzz_gpu.py 340 400 medium > $O/fuzz_medium.log 2>&1; echo "rc=$?"
tail -6 $O/fuzz_medium.log | cut -c1-700
