#!/bin/bash
# round-2 GPU call AD: XCD-major block order in the cell kernels -- full GPU suite, configs 2 / 3 / 1, general-path variants
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
run() { local nm=$1; shift
env "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -3 $O/bench_$nm.err; return 1; }
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], {k: p[k] for k in p if k in ("pcorrect","amd_diffusivities","tendencies","rhs","hydrostatic","fused_tracer_step","fused_tendency_step")})
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
run c2 $B --steps 200 --warmup 20 && run c3 $B --config 3 --steps 30 --warmup 6 && run c3_b $B --config 3 --steps 30 --warmup 6 && run ppb256 $B --topology PPB --steps 100 --warmup 10 && run pbb256 $B --topology PBB --steps 60 --warmup 10 && run c1 $B --config 1 --steps 2000 --warmup 20
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o trace --output-format csv -- python3 $R/bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --graph off > $O/trace_c3.log 2>&1
head -14 $O/trace_c3/trace_kernel_stats.csv | cut -c1-120
