#!/bin/bash
# tools/pmc_pass.sh <tag> <bench args...>: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs, as
# MI355X_MICROARCH.md prescribes) for ONE bench configuration; reductions land in gpurun_out/<tag>/:
#   kernel_stats.csv, traffic.json (per kernel: HBM bytes per launch with the gfx950 FETCH_SIZE correction, SQ_WAIT_ANY
#   fraction, VALU instructions per wave), bench.json (the bench line under the profiler).
# Copy what is to be judged into profiles/<round>/ as kernel_stats_<cfg>.csv / traffic_<cfg>.json (bench.py reads
# traffic_<cfg>.json, cfg = bench.config_tag()).
set -eo pipefail
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-also "$@" > $O/bench.json 2>$O/stats.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-also "$@" > /dev/null 2>$O/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-also "$@" > /dev/null 2>$O/write.err
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-also "$@" > /dev/null 2>$O/sq.err
echo "sq done"
cd $R
python3 tools/summarize_pmc.py $O/fetch $O/write $O/sq > $O/traffic.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/fetch $O/write $O/sq $O/stats
