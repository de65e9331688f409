#!/bin/bash
# Developer script (GPU box): instruction counters of the decoder kernel PER PHASE.  Builds the DIAG library in the box's copy of the tree and runs
# one SQ counter pass per phase exit (option fake_s: 32 = return behind the prologue + z, 128 = behind the first tanh layer, 64 = behind both,
# 0 = the whole kernel); differences between consecutive rows are the phases.  usage: bash tools/dev/pmc_phases.sh <tag>
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG=${1:-x}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/phases_$TAG; mkdir -p "$OUT"
cd "$R" && DIAG=1 bash iwae_amd/csrc/build.sh > "$OUT/build.log" 2>&1 || { tail "$OUT/build.log"; exit 1; }
cd /tmp && export TMPDIR=/tmp
for V in 32 128 64 0; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d "$OUT/v$V" -- python3 "$R/bench.py" --steps 12 --warmup 4 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval --opt fake_s=$V > "$OUT/v$V.log" 2>&1 || echo "pass $V failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for V in (32, 128, 64, 0):
    agg = collections.defaultdict(list); dur = []
    for fn in glob.glob("%s/v%d/*/*counter_collection.csv" % (out, V)):
        for r in csv.DictReader(open(fn)):
            if "bern_pipe_kernel<7, true, true, true>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for fn in glob.glob("%s/v%d/*/*kernel_trace.csv" % (out, V)):
        for r in csv.DictReader(open(fn)):
            if "bern_pipe_kernel<7, true, true, true>" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    dur.sort()
    print("exit %3d  alone %.1f us  " % (V, dur[len(dur) // 2] if dur else 0) + "  ".join("%s %.4g" % (k.replace("SQ_", ""), sum(v) / len(v)) for k, v in sorted(agg.items())))
PY
