#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace summary and the HBM-traffic counters of bench.py,
# each in its own pass (counters are never combined with other trace domains).  Results land in gpurun_out/prof_final/;
# tools/summarize_pmc.py turns them into profiles/*.json.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT -o pmc_$c --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES -d $OUT -o pmc_sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
ls $OUT
