# A/B of the pair kernel's line exchange on ONE box: GS_MILLER_TWIN=3 (DPP) against 2 (LDS slots)
# usage: bash tools/ab_exchange.sh "<bench args>" ...
export PYTHONPATH=$GRAFT_REPO_ROOT
for args in "$@"; do for rep in 1 2; do for tw in 3 2; do
  GS_MILLER_TWIN=$tw timeout -k 10 200 python bench.py $args --no-also --no-cpu --steps 4 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$args', 'DPP' if $tw==3 else 'LDS', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['kernels_ms'].items() if 'miller' in k})"
done; done; done
