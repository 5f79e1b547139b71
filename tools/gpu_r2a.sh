#!/bin/bash
# round-2 GPU call A: hardware micro-checks, GPU tests, bench (new WENO algebra; 1-Newton-step variant), kernel trace, VALU counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
./tools/micro_checks > $O/micro_checks.log 2>&1; echo "micro rc=$?" >> $O/micro_checks.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --lib clima-oceananigans.jl_amd/libocnhip_nr1.so > $O/bench_nr1.json 2> $O/bench_nr1.err
OCNHIP_TEST_LIB=$R/clima-oceananigans.jl_amd/libocnhip_nr1.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_gpu_nr1.log 2>&1; echo "pytest nr1 rc=$?" >> $O/pytest_gpu_nr1.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/pmc_sq -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
ls $O
