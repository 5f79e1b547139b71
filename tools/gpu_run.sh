#!/bin/bash
# One parametrised runner for everything that goes to the GPU box (replaces the per-call gpu_r*.sh scripts of rounds 1-2).
#   gpurun --timeout 900 -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'
# Every step writes gpurun_out/<tag>/<name>.log (+ .json for bench lines), prints its exit code and a short tail, and the
# script stops at the first step that was killed by its time limit (no GPU step is started after a hung one).
# Steps:
#   tests[=<pytest -k expression>]      python -m pytest tests -m gpu -x -q [-k ...]
#   bench:<name>[=<bench.py args>]      python bench.py <args> > <name>.json        (env assignments may precede the args)
#   trace:<name>[=<bench.py args>]      rocprofv3 --kernel-trace --stats -- python3 bench.py <args>   -> <name>_kernel_stats.csv
#   pmc:<name>[=<bench.py args>]        separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ set) of bench.py <args>
#   py:<name>=<script and args>         python <script and args>
#   pytrace:<name>=<script and args>    rocprofv3 --kernel-trace --stats -- python3 <script and args>  -> <name>_kernel_stats.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
LIMIT=${STEP_LIMIT:-500}
finish() { local frc=$1 fname=$2; echo "[$fname] rc=$frc"; if [ "$frc" = 124 ] || [ "$frc" = 137 ]; then echo "[$fname] hit its time limit: stopping"; exit 1; fi; }
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    print("  ms/step %.4f  value %.4g  dom %s %.4f ms frac %.3f  step_frac %s  div %s" % (
        d["ms_per_step"], d["value"], r.get("kernel"), r.get("avg_launch_ms", float("nan")), r.get("frac", float("nan")),
        d.get("step_roofline", {}).get("frac_of_hbm_peak"), d.get("max_abs_divergence")))
    print("  phases", d.get("phases_ms_warmup"))
except Exception as e:
    print("  (no bench line:", e, ")")
PY
}
for step in "$@"; do
  kind=${step%%[:=]*}
  rest=${step#"$kind"}
  name=${rest#:}; name=${name%%=*}
  args=""; case "$step" in *=*) args=${step#*=};; esac
  case "$kind" in
    tests)
      if [ -n "$args" ]; then timeout -k 10 $LIMIT python -m pytest tests -m gpu -x -q -k "$args" > "$O/pytest.log" 2>&1
      else timeout -k 10 ${TESTS_LIMIT:-1100} python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; fi
      rc=$?; tail -4 "$O/pytest.log" | cut -c1-400; finish $rc tests;;
    bench)
      # leading VAR=value words become the environment of the run
      envs=(); set -- $args; while [ $# -gt 0 ] && [[ "$1" == *=* ]] && [[ "$1" != -* ]]; do envs+=("$1"); shift; done
      env "${envs[@]}" timeout -k 10 $LIMIT python bench.py "$@" > "$O/$name.json" 2> "$O/$name.err"; rc=$?
      summ "$O/$name.json"; finish $rc "bench:$name";;
    trace)
      ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 $LIMIT rocprofv3 --kernel-trace --stats -d "$O/$name" -o trace --output-format csv -- \
          python3 "$R/bench.py" --no-cpu-baseline $args > "$O/$name.log" 2>&1 ); rc=$?
      f=$(find "$O/$name" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv" && head -12 "$f" | cut -c1-200
      finish $rc "trace:$name";;
    pmc)
      for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
        tagc=$(echo $set | cut -d' ' -f1)
        ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 $LIMIT rocprofv3 --pmc $set -d "$O/${name}_$tagc" -o pmc --output-format csv -- \
            python3 "$R/bench.py" --no-cpu-baseline --steps 3 --warmup 1 $args > "$O/${name}_$tagc.log" 2>&1 ); rc=$?
        finish $rc "pmc:$name:$tagc"
      done
      python tools/summarize_pmc.py "$O" "$name" > "$O/${name}_pmc_summary.json" 2> "$O/${name}_pmc_summary.err" || true
      head -c 1500 "$O/${name}_pmc_summary.json";;
    pytrace)
      set -- $args; script=$1; shift
      ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 $LIMIT rocprofv3 --kernel-trace --stats -d "$O/$name" -o trace --output-format csv -- \
          python3 "$R/$script" "$@" > "$O/$name.log" 2>&1 ); rc=$?
      f=$(find "$O/$name" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv" && head -14 "$f" | cut -c1-200
      finish $rc "pytrace:$name";;
    py)
      timeout -k 10 $LIMIT python $args > "$O/$name.log" 2>&1; rc=$?; tail -8 "$O/$name.log" | cut -c1-400; finish $rc "py:$name";;
    *) echo "unknown step $step"; exit 2;;
  esac
done
ls "$O" | head -50
