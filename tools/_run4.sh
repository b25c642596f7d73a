cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
GS_PIPE_TRACE=1 timeout -k 10 200 python tools/host_path_rate.py 16 1 2> gpurun_out/r3c/pipe_trace16.txt | tail -1
tail -45 gpurun_out/r3c/pipe_trace16.txt
for sp in 1 2; do GS_COPY_SPLIT=$sp timeout -k 10 200 python tools/host_path_rate.py 16 3 2>/dev/null | tail -1; done
GS_COPY_THREADS=8 timeout -k 10 200 python tools/host_path_rate.py 16 3 2>/dev/null | tail -1
timeout -k 10 200 python tools/host_path_rate.py 12 10 2>/dev/null | tail -1
timeout -k 10 200 python tools/host_path_rate.py 14 5 2>/dev/null | tail -1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3c/pytest_all.log 2>&1; tail -5 gpurun_out/r3c/pytest_all.log
