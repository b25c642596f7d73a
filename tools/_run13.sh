cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout -k 10 600 python bench.py --no-cpu > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err || tail -20 gpurun_out/r3i/bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3i/bench.json'))
a=d['roofline']['alu']
print('main', round(d['value']), round(d['ms_per_step'],1), d['roofline']['kernel'], 'alu', round(a['frac'],3), 'exec', round(a['executed_mad_frac'],3), 'dom exec', round(a['dominant']['executed_mad_frac'],3), {k:round(v,1) for k,v in d['roofline']['kernels_ms'].items() if v>3})
for k,v in d['also'].items():
    print(k, round(v['value']), round(v['ms_per_step'],1), v.get('ratio_to_device_resident'))
PY
timeout -k 10 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fallback_builds.py > gpurun_out/r3i/pytest_all.log 2>&1; tail -4 gpurun_out/r3i/pytest_all.log
