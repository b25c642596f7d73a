cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/mixed_rate.py 16 2>/dev/null | sed -e 's/2^16 mixed 50\/25\/25: //'; }
run A=1
run HSA_NO_SCRATCH_THREAD_LIMITER=1
run HSA_SCRATCH_MEM=17179869184
run HSA_SCRATCH_SINGLE_LIMIT=8589934592
run HSA_NO_SCRATCH_RECLAIM=1
run HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
