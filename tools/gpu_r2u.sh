#!/bin/bash
# round-2 GPU call U: k_amd_all with compile-time tracer count; hydrostatic chunk sizes; AMD parity
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2u
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "amd or ppb_ or regr_" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
run() { local nm=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 30 --warmup 6 > $O/bench_$nm.json 2> $O/bench_$nm.err
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], "amd", p.get("amd_diffusivities"), "hydro", p.get("hydrostatic"), d["max_abs_divergence"])
PY
}
run c3 A=1 && run c3_ch16 OCNHIP_HYDRO_CH=16 && run c3_ch32 OCNHIP_HYDRO_CH=32 && run c3_b A=1
