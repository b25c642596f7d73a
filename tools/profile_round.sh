#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r2
# -> gpurun_out/<tag>/: the default bench line, bench lines of the other configurations, rocprofv3 kernel stats and the
#    separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*) of the 2^16 and 2^12 PPE configurations reduced by
#    tools/summarize_pmc.py, the large-arity latencies.  Copy what is to be judged into profiles/<tag>/.
set -eo pipefail
TAG=${1:-r3}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
tools/pmc_pass.sh $TAG/pmc_2p16_ppe --log2n 16
echo "pmc 2^16 done"
tools/pmc_pass.sh $TAG/pmc_2p12_ppe --log2n 12
echo "pmc 2^12 done"
python3 bench.py > $O/bench_default.json
echo "bench default done"
for args in "--log2n 15" "--log2n 17" "--log2n 14" "--log2n 16 --mode rlc" "--log2n 14 --type 1" "--log2n 14 --type 2" "--log2n 14 --type 3"; do
  name=$(echo "$args" | tr -d ' -')
  python3 bench.py --no-cpu --no-also --steps 3 --warmup 1 $args > $O/bench_$name.json
  echo "done $name"
done
python3 tools/large_arity_rate.py > $O/large_arity_334.json
for n in 16 14 12; do python3 tools/host_path_rate.py $n 4 2>/dev/null | tail -1; done > $O/host_path_rate.txt
python3 tools/mixed_rate.py 10 12 14 16 2>/dev/null > $O/mixed_merge.txt
ls -la $O
