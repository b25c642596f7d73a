cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
echo "== default"; timeout -k 10 300 python tools/_exp_mixed3.py m12 p12 m16 p16 2>/dev/null
echo "== GS_OVERLAP=0"; GS_OVERLAP=0 timeout -k 10 300 python tools/_exp_mixed3.py m12 p12 m16 p16 2>/dev/null
echo "== HSA_SCRATCH_MEM=32G"; HSA_SCRATCH_MEM=34359738368 timeout -k 10 300 python tools/_exp_mixed3.py m12 p12 m16 p16 2>/dev/null
echo "== HSA_NO_SCRATCH_THREAD_LIMITER=1"; HSA_NO_SCRATCH_THREAD_LIMITER=1 timeout -k 10 300 python tools/_exp_mixed3.py m12 p12 m16 p16 2>/dev/null
