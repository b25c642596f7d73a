#!/bin/bash
# tools/pmc_diag.sh <tag> <bench args...>: wider SQ / GRBM counter passes of ONE bench configuration (one pass per
# counter group: rocprofv3 PMC slots, MI355X_MICROARCH.md), reduced per kernel by tools/summarize_diag.py into
# gpurun_out/<tag>/diag.json: where the wave cycles of each kernel go (issue, s_waitcnt parking, issue stalls), the
# instruction mix, the instruction-cache behaviour and the effective clock (GRBM_GUI_ACTIVE / 8 / duration).
set -eo pipefail
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
pass() {  # name, counters...
  local n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-also $BARGS > /dev/null 2>$O/$n.err || { echo "pass $n failed"; tail -5 $O/$n.err; }
  echo "pass $n done"
}
BARGS="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_BRANCH
pass sq3 SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
pass sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
python3 tools/summarize_diag.py $O > $O/diag.json
for n in sq1 sq2 sq3 sqc grbm; do rm -rf $O/$n; done
