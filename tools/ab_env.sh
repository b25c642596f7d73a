#!/bin/bash
# A/B of planner overrides on ONE box: tools/ab_env.sh OUTDIR "bench args" "ENV1=.. ENV2=.." "..." ("-" = no override)
OUT=$1; ARGS=$2; shift 2
mkdir -p "$OUT"
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" = "-" ]; then e=""; fi
  env $e timeout -k 10 300 python3 bench.py --no-also --no-cpu $ARGS > "$OUT/env_$i.json" 2> "$OUT/env_$i.err" || { echo "$e FAILED"; tail -5 "$OUT/env_$i.err"; exit 1; }
  python3 - "$OUT/env_$i.json" "${e:-planned}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d['roofline']['kernels_ms']
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],1), {a:round(b,1) for a,b in k.items() if 'var' in a})
PY
done
