#!/bin/bash
# round-2 GPU call AE: fused z stage with XCD-major column tiles -- transform tests, solve phase at several sizes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_fft_sizes.py tests/test_parity_gpu.py -m gpu -x -q -k "transform or headline or full_size or medium" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
run() { local nm=$1; shift
env "$@" > $O/bench_$nm.json 2> $O/bench_$nm.err || { echo "bench $nm failed"; tail -3 $O/bench_$nm.err; return 1; }
python - <<PY
import json
d=json.load(open("$O/bench_$nm.json")); p=d["phases_ms_warmup"]; print("$nm ms/step %.4f" % d["ms_per_step"], "fwd", p.get("fft_forward"), "solve", p.get("spectral_solve"), "bwd", p.get("fft_backward"), "proj", p.get("pcorrect"))
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20"
run s256 $B && run s512x128 $B --size 512 512 128 --steps 60 && run s128 $B --size 128 128 128 && run s256x512 $B --size 256 256 512 --steps 60 && run s256_b $B
