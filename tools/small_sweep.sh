export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t
run() { GS_PLAN_TRACE=1 timeout -k 10 150 python bench.py "$@" --no-also --no-cpu --steps 3 --warmup 1 2>gpurun_out/r3t/e.txt | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['kernels_ms'].items() if 'miller' in k or 'final' in k})"; grep "\[plan\]" gpurun_out/r3t/e.txt | sort | uniq | cut -c1-220; }

for l in 6 8 10 11 12 13 14 16; do run --log2n $l; done
run --mixed --log2n 12
run --mixed --log2n 10
run --curve 1 --log2n 16
run --curve 1 --log2n 12
run --type 2 --log2n 12
