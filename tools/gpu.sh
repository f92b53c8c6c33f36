#!/bin/bash
# One GPU-box session (run through gpurun): a tag and a list of steps, output under gpurun_out/<tag>/.
#
#   gpurun -- bash tools/gpu.sh <tag> <step> [<step> ...]
#
# steps
#   tests[:<pytest -k expression>]   the GPU test suite (or a part of it)          -> gpu_tests.log, parity_errors.json
#   smoke                            __graft_entry__.smoke()                        -> smoke.log
#   bench:<label>:<bench.py args>    one bench line (args separated by spaces)      -> bench_lines.jsonl
#   profile:<config>                 tools/profile_gpu.sh (rocprofv3 stats + HBM PMC passes)
#   pmc:<config>                     tools/profile_pmc.sh (SQ counters)
#   ab:<bench args>:<variant>,<variant>,...   alternate library variants (main = in-tree, else scratch/<name>/), two rounds
#   py:<script> [args]               python <script> under tools/ or scratch/, output -> <script>.log
#   lib:<variant>                    later steps load scratch/<variant>/libdpc_render.so (tools/build_variant.sh); lib:main = in-tree
# A step that fails stops the session (steps are joined with &&: nothing runs on a GPU a failed step may have left bad).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/$tag; mkdir -p $OUT
summ() { python3 - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try:
        d = json.loads(l)
    except Exception:
        print("unparsed:", l[:160]); continue
    j = d.get("line", d)
    ks = {k.replace("k_", ""): round(x, 2) for k, x in j.get("kernels_us", {}).items()}
    print("%-70s %10.0f /s  %9.2f us  %s" % (d.get("label", "")[:70], j["value"], 1e3 * j["ms_per_step"], ks))
PY
}
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  case $kind in
    tests)
      if [ -n "$rest" ]; then timeout -k 10 1100 python -m pytest tests -q -m gpu -k "$rest" > $OUT/gpu_tests.log 2>&1
      else timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; fi
      rc=$?; tail -4 $OUT/gpu_tests.log; cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
      [ $rc -ne 0 ] && { grep -n "^FAILED\|^ERROR\|Error\|assert" $OUT/gpu_tests.log | tail -30; echo "tests exit=$rc"; exit $rc; } ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; tail -2 $OUT/smoke.log
      [ $rc -ne 0 ] && { echo "smoke exit=$rc"; exit $rc; } ;;
    bench)
      label=${rest%%:*}; args=${rest#*:}
      line=$(timeout -k 10 400 python bench.py $args 2>>$OUT/bench.err | tail -1); rc=$?
      [ $rc -ne 0 ] && { echo "bench '$label' exit=$rc"; tail -5 $OUT/bench.err; exit $rc; }
      echo "{\"label\": \"$label\", \"line\": $line}" >> $OUT/bench_lines.jsonl; summ $OUT/bench_lines.jsonl | tail -1 ;;
    profile)
      bash tools/profile_gpu.sh $rest > $OUT/profile_$rest.log 2>&1; rc=$?; tail -3 $OUT/profile_$rest.log
      [ $rc -ne 0 ] && { echo "profile exit=$rc"; exit $rc; } ;;
    pmc)
      bash tools/profile_pmc.sh $rest > $OUT/pmc_$rest.log 2>&1; rc=$?; tail -8 $OUT/pmc_$rest.log
      [ $rc -ne 0 ] && { echo "pmc exit=$rc"; exit $rc; } ;;
    ab)
      args=${rest%%:*}; variants=${rest#*:}; f=$OUT/ab_$(echo "$args" | tr -c 'a-zA-Z0-9' '_').jsonl
      for rnd in 1 2; do
        for v in ${variants//,/ }; do
          lib=""; [ "$v" != main ] && lib="$PWD/scratch/$v/libdpc_render.so"
          line=$(DPC_RENDER_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>>$OUT/bench.err | tail -1); rc=$?
          [ $rc -ne 0 ] && { echo "ab variant $v exit=$rc"; tail -5 $OUT/bench.err; exit $rc; }
          echo "{\"label\": \"$v ($args)\", \"line\": $line}" >> $f
        done
      done
      summ $f ;;
    py)
      script=${rest%% *}; timeout -k 10 900 python $rest > $OUT/$(basename $script).log 2>&1; rc=$?; tail -15 $OUT/$(basename $script).log
      [ $rc -ne 0 ] && { echo "py exit=$rc"; exit $rc; } ;;
    lib)   # lib:<variant>  -- every later step of the session loads scratch/<variant>/libdpc_render.so (lib:main = the in-tree one)
      if [ "$rest" = main ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB="$PWD/scratch/$rest/libdpc_render.so"; fi
      echo "library: ${DPC_RENDER_LIB:-in-tree}" ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
exit 0
