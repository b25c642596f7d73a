cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 300 python tools/mixed_rate.py 10 12 14 16 2>/dev/null | tee gpurun_out/r3e/mixed_merge.txt
for n in 16 14 12; do timeout -k 10 200 python tools/host_path_rate.py $n 4 2>/dev/null | tail -1; done | tee gpurun_out/r3e/host_path_rate.txt
tools/pmc_pass.sh r3e/pmc_2p16_ppe --log2n 16 > gpurun_out/r3e/pmc.log 2>&1; tail -3 gpurun_out/r3e/pmc.log
python3 -c "
import json
d=json.load(open('gpurun_out/r3e/pmc_2p16_ppe/traffic.json'))
for k,v in d.items():
    if 'miller' in k or 'final' in k or 'var_multi' in k: print(k[:80], round(v['hbm_bytes_per_launch_corrected']/1e9,1),'GB', v.get('wait_any_frac'), v.get('valu_insts_per_wave'))
"
