cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
tools/ab_bench.sh gpurun_out/r3h "--steps 6 --warmup 2" main lineorder main lineorder | tee gpurun_out/r3h/ab_lineorder.txt
GS_AMD_LIB=$PWD/groth_sahai_rs_amd/lib/var/lineorder.so timeout -k 10 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_fullsize.py -q -x -k "bls12_381 and not bn254" > gpurun_out/r3h/pytest_lineorder.log 2>&1; tail -3 gpurun_out/r3h/pytest_lineorder.log
for n in 16 12; do timeout -k 10 200 python tools/host_path_rate.py $n 4 2>/dev/null | tail -1; done | tee gpurun_out/r3h/host_path_rate.txt
timeout -k 10 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fallback_builds.py > gpurun_out/r3h/pytest_all.log 2>&1; tail -4 gpurun_out/r3h/pytest_all.log
