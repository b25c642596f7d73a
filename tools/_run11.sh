cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout -k 10 600 python -m pytest tests/test_gpu_variants.py -q -x -k "tab8 or validated" > gpurun_out/r3g/pytest_tab8.log 2>&1; tail -3 gpurun_out/r3g/pytest_tab8.log
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -q -x -k "334 or arity or wide or fold" > gpurun_out/r3g/pytest_334.log 2>&1; tail -3 gpurun_out/r3g/pytest_334.log
timeout -k 10 400 python tools/large_arity_rate.py > gpurun_out/r3g/large_arity_334.json 2>gpurun_out/r3g/la.err; python3 -c "
import json;d=json.load(open('gpurun_out/r3g/large_arity_334.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, 'prove %.1f verify %.1f gamma-msm %.1f' % (v['prove_ms'], v['verify_ms'], v['gamma_msm_ms']), {a:b for a,b in v['kernels_ms'].items() if 'vg1' in a or 'miller' in a or 'final' in a})
"
for n in 16 12; do timeout -k 10 200 python tools/host_path_rate.py $n 4 2>/dev/null | tail -1; done | tee gpurun_out/r3g/host_path_rate.txt
timeout -k 10 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fallback_builds.py > gpurun_out/r3g/pytest_all.log 2>&1; tail -4 gpurun_out/r3g/pytest_all.log
