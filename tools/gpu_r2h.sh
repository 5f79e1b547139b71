#!/bin/bash
# round-2 GPU call H: record profiles (kernel trace, traffic and VALU counters) + bench lines of the secondary configs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("$name", "ms/step %.4f dom %s %.4f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]), "step_frac %.3f" % d["step_roofline"]["frac_of_hbm_peak"], d["max_abs_divergence"], d["phases_ms_warmup"])
except Exception as e: print("$name ERR", e)
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100"
run default_full timeout -k 10 400 python bench.py
run s128 $B --size 128 128 128
run s512x32_forced OCNHIP_FORCE_DIST=1 $B --size 512 512 32
run s512x128 $B --size 512 512 128 --steps 40
run c3 $B --config 3 --steps 20 --warmup 5
run rk3 $B --stepper RK3 --steps 40
run tr1 $B --tracers 1
run c1 $B --config 1
OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_1gpu_shm.json 2> $O/bench_2ranks_1gpu_shm.err; echo "2-rank rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/trace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o trace --output-format csv -- python3 $R/bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_c3.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/pmc_sq -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_fft_sizes.py -m gpu -x -q -k "amd or config3 or transform or forced" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sel.log
tail -3 $O/pytest_sel.log
ls $O
