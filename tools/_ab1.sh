set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_variants.py -x -q -k "pair" > gpurun_out/r3a/pytest_pair.log 2>&1 || { tail -30 gpurun_out/r3a/pytest_pair.log; exit 1; }
tail -3 gpurun_out/r3a/pytest_pair.log
tools/ab_env.sh gpurun_out/r3a "--steps 6 --warmup 2" "GS_MILLER_TWIN=1" "GS_MILLER_TWIN=2" "GS_MILLER_TWIN=3" "GS_MILLER_TWIN=1" "GS_MILLER_TWIN=2" | tee gpurun_out/r3a/ab.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/env_*.json')):
    d=json.load(open(f)); print(f, round(d['value']), {k:round(v,1) for k,v in d['roofline']['kernels_ms'].items() if 'miller' in k})
PY
