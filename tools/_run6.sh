cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3d
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3d/mtrace -- python3 $R/tools/_exp_mixed5.py 12 > /dev/null 2> $R/gpurun_out/r3d/mtrace.err
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3d/mtrace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'k_seg' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last third = one full step
n = len(rows) // 3
step = rows[-n:]
t0 = int(step[0]['Start_Timestamp'])
for r in step:
    nm = r['Kernel_Name'].split('k_seg<gs::')[1][:40]
    print('%8.2f ms  +%6.2f ms  grid %7s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'), nm))
print('step total %.2f ms, %d launches' % ((int(step[-1]['End_Timestamp']) - t0) / 1e6, len(step)))
PY
