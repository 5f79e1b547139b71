#!/bin/bash
# round-2 GPU call V: column-march AMD predictor kernel -- parity and config 3 with segment lengths 0 (old kernel), 8, 16, 32, 64
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2v
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "amd or config3" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
run() { local nm=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 30 --warmup 6 > $O/bench_$nm.json 2> $O/bench_$nm.err
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], "amd", p.get("amd_diffusivities"), d["max_abs_divergence"])
PY
}
run seg0 OCNHIP_AMD_SEG=0 && run seg8 OCNHIP_AMD_SEG=8 && run seg16 OCNHIP_AMD_SEG=16 && run seg32 OCNHIP_AMD_SEG=32 && run seg64 OCNHIP_AMD_SEG=64 && run seg128 OCNHIP_AMD_SEG=128 && run dflt A=1
