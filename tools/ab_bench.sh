#!/bin/bash
# A/B on ONE box: tools/ab_bench.sh OUTDIR "bench args" main var1 var2 ...   ("main" = the product library)
# Each variant: lib/var/NAME.so built by tools/build_variant.sh.  Prints ms/step and the per-kernel times.
OUT=$1; ARGS=$2; shift 2
mkdir -p "$OUT"
for v in "$@"; do
  if [ "$v" = main ]; then unset GS_AMD_LIB; else export GS_AMD_LIB=$PWD/groth_sahai_rs_amd/lib/var/$v.so; fi
  timeout -k 10 300 python3 bench.py --no-also --no-cpu $ARGS > "$OUT/ab_$v.json" 2> "$OUT/ab_$v.err" || { echo "$v FAILED"; tail -5 "$OUT/ab_$v.err"; exit 1; }
  python3 - "$OUT/ab_$v.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d['roofline']['kernels_ms']
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],1), {a:round(b,1) for a,b in k.items() if b>1.5})
PY
done
