cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
echo "== sequence"; timeout -k 10 300 python tools/_exp_mixed3.py m12 p12 m16 p16 p12 p16 2>/dev/null
timeout -k 10 400 python tools/mixed_rate.py 10 12 14 16 2>/dev/null | tee gpurun_out/r3f/mixed_merge.txt
timeout -k 10 600 python bench.py --no-cpu > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err || tail -20 gpurun_out/r3f/bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3f/bench.json'))
print('main', round(d['value']), round(d['ms_per_step'],1), d['roofline']['kernel'], {k:round(v,1) for k,v in d['roofline']['kernels_ms'].items() if v>3})
for k,v in d['also'].items():
    print(k, round(v['value']), round(v['ms_per_step'],1), v.get('ratio_to_device_resident'))
PY
timeout -k 10 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fallback_builds.py > gpurun_out/r3f/pytest_all.log 2>&1; tail -4 gpurun_out/r3f/pytest_all.log
