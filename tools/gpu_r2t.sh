#!/bin/bash
# round-2 GPU call T: tracer G^n read / cleared only in the boundary levels (shell mode) -- parity + config 3, kernel trace of config 3
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2t
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_known_answers.py tests/test_model_contracts.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for nm in c3 c3_b; do
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 30 --warmup 6 > $O/bench_$nm.json 2> $O/bench_$nm.err
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], "tracer", p.get("fused_tracer_step"), "tend", p.get("tendencies"), "solve", p.get("spectral_solve"), d["max_abs_divergence"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o trace --output-format csv -- python3 $R/bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --graph off > $O/trace_c3.log 2>&1
head -12 $O/trace_c3/trace_kernel_stats.csv | cut -c1-160
