#!/bin/bash
# Developer script (GPU box): SQ counter passes over the bench loop, per-kernel per-launch averages.
# usage: bash tools/dev/pmc_sq.sh <tag> [kernel-name-substring ...]
TAG=${1:-x}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-llh-eval"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]; pats = sys.argv[2:] or ["bern_pipe_kernel", "out_bwd", "dense_kernel<0, 7", "wgradp_kernel<16, true"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if not any(p in k for p in pats): continue
    print(k[:90])
    for n in sorted(c): print("    %-34s %14.4g" % (n, sum(c[n]) / len(c[n])))
PY
