set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_hostpath.py tests/test_gpu_mixed.py tests/test_gpu_mirror.py -x -q > gpurun_out/r3b/pytest_new.log 2>&1 || { tail -60 gpurun_out/r3b/pytest_new.log; exit 1; }
tail -3 gpurun_out/r3b/pytest_new.log
timeout -k 10 600 python bench.py --steps 6 --warmup 1 --no-cpu > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err || { tail -30 gpurun_out/r3b/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print('main', round(d['value']), d['ms_per_step'], d['roofline']['kernel'], d['roofline']['alu'].get('executed_mad_frac'), d['roofline']['alu'].get('frac'))
for k,v in d['also'].items():
    print(k, round(v['value']), round(v['ms_per_step'],1), v.get('ratio_to_device_resident'), v.get('dominant_kernel'))
PY
