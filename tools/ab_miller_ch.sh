# A/B of the Miller lane length on ONE box: bench lines at 2^LOG2N with the planner's choice and with GS_MILLER_CH forced
# usage: bash tools/ab_miller_ch.sh "<log2n list>" "<ch list, 0 = planned>" [extra bench args]
export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t
L="$1"; CH="$2"; shift 2
for l in $L; do for rep in 1 2; do for ch in $CH; do
  if [ $ch = 0 ]; then unset GS_MILLER_CH; else export GS_MILLER_CH=$ch; fi
  GS_PLAN_TRACE=1 timeout -k 10 200 python bench.py --log2n $l "$@" --no-also --no-cpu --steps 4 --warmup 1 2>gpurun_out/r3t/e.txt | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('log2n $l ch $ch:', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['kernels_ms'].items() if 'miller' in k or 'final' in k})"
  grep "\[plan\]" gpurun_out/r3t/e.txt | sort | uniq | sed 's/.*budget/   budget/' | cut -c1-150
done; done; done
