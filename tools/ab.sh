#!/bin/bash
# A/B of library variants on the GPU box: tools/ab.sh <out-name> <bench args...> -- <variant> [<variant> ...]
# variant = "main" (in-tree library) or a scratch/<name> build; each runs bench.py twice, interleaved.
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
out=gpurun_out/r03/$1; shift
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
: > $out
for rnd in 1 2; do
  for v in "$@"; do
    lib=""; [ "$v" != main ] && lib="$PWD/scratch/$v/libdpc_render.so"
    line=$(DPC_RENDER_LIB=$lib python bench.py --no-cpu-baseline "${args[@]}" 2>/dev/null | tail -1)
    echo "$v $line" >> $out
  done
done
python - $out <<'PY'
import json,sys
for l in open(sys.argv[1]):
    v,j=l.split(' ',1)
    try: d=json.loads(j)
    except Exception as e: print(v,'FAILED',j[:200]); continue
    print("%-10s %8.0f clouds/s  wall %.2f us  dev %.2f us  %s  median %s" % (v, d['value'], 1e3*d['ms_per_step'], 1e3*d['roofline_step']['device_ms_per_step'], {k.replace('k_',''):round(x,2) for k,x in d['kernels_us'].items()}, round(d.get('step_us',{}).get('median',0),2)))
PY
