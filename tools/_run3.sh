cd $GRAFT_REPO_ROOT
for t in 2 4 8 12; do GS_COPY_THREADS=$t timeout -k 10 200 python tools/host_path_rate.py 16 3 || exit 1; done
for t in 2 4 8; do GS_COPY_THREADS=$t timeout -k 10 200 python tools/host_path_rate.py 12 10 || exit 1; done
