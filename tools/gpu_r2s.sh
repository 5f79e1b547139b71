#!/bin/bash
# round-2 GPU call S: Thomas kernel with loads one chunk ahead and the mean subtraction as its own launch
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2s
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_properties_lib.py -m gpu -x -q -k "ppb_ or regr_ or config3 or poisson or stretched or incompressible" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for nm in c3 c3_b; do
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 30 --warmup 6 > $O/bench_$nm.json 2> $O/bench_$nm.err
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], "solve", p.get("spectral_solve"), "hydro", p.get("hydrostatic"), "amd", p.get("amd_diffusivities"), d["max_abs_divergence"])
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline --topology PPB --steps 100 --warmup 10 > $O/bench_ppb.json 2> $O/bench_ppb.err
python - <<PY
import json
d=json.load(open("$O/bench_ppb.json")); p=d["phases_ms_warmup"]; print("ppb256 ms/step %.4f" % d["ms_per_step"], "solve", p.get("spectral_solve"), d["max_abs_divergence"])
PY
