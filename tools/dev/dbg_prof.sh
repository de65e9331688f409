#!/bin/bash
# Developer script (GPU box): timing + rocprof per-kernel stats + one-step timeline for a config.
B=${1:-1024}; K=${2:-50}; L=${3:-1}; TAG=${4:-run}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/tools/dev/dbg_time.py $B $K $L 300
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/dev/dbg_time.py $B $K $L 50 > $R/gpurun_out/prof_$TAG.log 2>&1 || echo "rocprof failed"
python3 - <<PY
import csv, glob
fs = glob.glob('$R/gpurun_out/prof_$TAG/*/*kernel_trace.csv')
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[a]['Start_Timestamp']); tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp']); tot += e - s
    print("%8.1f us  dur %7.1f  grid %-14s lds %-6s vgpr %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r['Grid_Size_X'] + 'x' + r['Grid_Size_Y'] + 'x' + r['Grid_Size_Z'], r['LDS_Block_Size'], r['VGPR_Count'], r['Kernel_Name'][:50]))
print("sum %.1f us span %.1f us" % (tot / 1e3, (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3))
PY
