#!/bin/bash
# Developer script (GPU box): PMC passes (own runs, --pmc only) for the hot kernels.
B=${1:-1024}; K=${2:-50}; L=${3:-1}; TAG=${4:-pmc}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"
P3="SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/${TAG}_p$i -- python3 $R/tools/dev/dbg_time.py $B $K $L 10 > $R/gpurun_out/${TAG}_p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    for f in glob.glob('$R/gpurun_out/${TAG}_p%d/*/*counter_collection.csv' % i):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted(agg, key=lambda n: -sum(agg[n].get('SQ_BUSY_CYCLES', [0])))
for n in names[:12]:
    c = {k: sum(v) / len(v) for k, v in agg[n].items()}
    print("== %s (dispatches %d)" % (n, len(agg[n].get('SQ_WAVE_CYCLES', []))))
    print("   " + "  ".join("%s=%.3g" % (k.replace('SQ_', ''), v) for k, v in sorted(c.items())))
PY
