cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout -k 10 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fallback_builds.py > gpurun_out/r3d/pytest_all.log 2>&1; tail -8 gpurun_out/r3d/pytest_all.log
GS_PIPE_TRACE=1 timeout -k 10 200 python tools/host_path_rate.py 16 1 2> gpurun_out/r3d/pipe_trace16.txt | tail -1
head -24 gpurun_out/r3d/pipe_trace16.txt | grep pipe
timeout -k 10 200 python tools/host_path_rate.py 16 3 2>/dev/null | tail -1
timeout -k 10 200 python tools/host_path_rate.py 12 10 2>/dev/null | tail -1
timeout -k 10 200 python tools/_exp_mixed3.py m12 p12 m16 p16 2>/dev/null
