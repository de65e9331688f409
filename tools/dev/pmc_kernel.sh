#!/bin/bash
# Developer script (GPU box): SQ / TCC counter passes over the bench loop WITH a kernel trace in the same run (counter collection
# serialises the dispatches, so the traced durations are each kernel ALONE on the machine), per-kernel per-launch averages.
# usage: bash tools/dev/pmc_kernel.sh <tag> [kernel-name-substring ...]      env: BENCH_ARGS (extra bench.py arguments), PK_CMD (profile another command)
# PK_CMD must START with the interpreter / program itself (python3 ...): no env, bash -c, taskset or other hop after rocprofv3's `--`
# (the profiler's preloaded library has initialised the GPU by then; a re-exec from there takes the box down).
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG=${1:-x}; shift || true
BENCH_ARGS=${BENCH_ARGS:-}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pk_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD=${PK_CMD:-"python3 $R/bench.py --steps 30 --warmup 10 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval $BENCH_ARGS"}      # (PK_CMD: another program to profile, e.g. "python3 $GRAFT_REPO_ROOT/tools/dev/eval_only.py fp32 400")
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]; pats = sys.argv[2:] or ["wgradp_kernel<16, true", "dec_bwd", "bern_pipe_kernel<7, true"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for fn in glob.glob(out + "/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    if not any(p in k for p in pats): continue
    d = sorted(dur.get(k, [0.0]))
    print("%s   alone: median %.1f us (n=%d)" % (k[:90], d[len(d) // 2], len(d)))
    for n in sorted(c): print("    %-34s %14.4g" % (n, sum(c[n]) / len(c[n])))
PY
