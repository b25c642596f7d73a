#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r1
# -> gpurun_out/<tag>/: bench jsons, rocprofv3 kernel stats, and the three separate PMC passes
#    (FETCH_SIZE, WRITE_SIZE, SQ_*) reduced by tools/summarize_pmc.py.  Copy what is to be judged into profiles/<tag>/.
set -eo pipefail
TAG=${1:-r1}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench_default.json
for args in "--log2n 14" "--log2n 16" "--log2n 16 --mode rlc" "--log2n 16 --mixed" "--log2n 16 --curve 1" "--log2n 14 --type 1" "--log2n 14 --type 2" "--log2n 14 --type 3"; do
  name=$(echo "$args" | tr -d ' -')
  python3 bench.py --no-cpu --steps 3 --warmup 1 $args > $O/bench_$name.json
  echo "done $name"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/bench_under_rocprof.json 2>$O/stats.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>$O/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>$O/write.err
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>$O/sq.err
echo "sq done"
cd $R
python3 tools/summarize_pmc.py $O/fetch $O/write $O/sq > $O/traffic_2p12.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_2p12_kernel_stats.csv
# the raw traces are large; keep the reductions only
rm -rf $O/fetch $O/write $O/sq $O/stats
ls -la $O
