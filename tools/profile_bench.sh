#!/bin/bash
# Profiles a bench command on the GPU box and writes summaries under gpurun_out/profile_<tag>/
# (copy the ones to keep into profiles/).  Kernel-trace/stats and PMC runs are separate invocations.
# usage: tools/profile_bench.sh <tag> [config] [quick]      quick: kernel trace + FETCH/WRITE passes only
TAG=${1:-r02}
CONFIG=${2:-c1}
QUICK=${3:-}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=60
WARM=10
CMD="python3 $R/bench.py --steps $STEPS --warmup $WARM --settle 0 --no-kernel-times --config $CONFIG --no-cpu-baseline --no-llh-eval $BENCH_ARGS"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace run failed"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
if [ -z "$QUICK" ]; then
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_mfma -- $CMD > $OUT/pmc_mfma.log 2>&1 || echo "pmc mfma failed"
fi
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
nsteps = $STEPS + $WARM + 1          # + the forward bench.py runs after the timed region
rows = list(csv.DictReader(open(glob.glob(out + "/trace/*/*kernel_stats.csv")[0])))
with open(out + "/kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows: w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mfma"):
    for fn in glob.glob(out + "/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {}
for k, c in agg.items():
    summ[k] = {n: sum(v) / len(v) for n, v in c.items()}
    summ[k]["dispatches"] = len(next(iter(c.values())))
json.dump(summ, open(out + "/pmc_per_launch.json", "w"), indent=1, sort_keys=True)
# gfx950: FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact
stat = {r["Name"]: r for r in rows}
kern, step_bytes, lines = [], 0.0, []
for k, v in sorted(summ.items(), key=lambda kv: -float(stat.get(kv[0], {"TotalDurationNs": 0})["TotalDurationNs"])):
    if k not in stat: continue
    r = stat[k]
    per_step = int(r["Calls"]) / float($STEPS + $WARM)
    rd = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0; wr = v.get("WRITE_SIZE", 0.0) * 1024.0
    us = float(r["AverageNs"]) / 1e3
    kern.append({"kernel": k, "launches_per_step": round(per_step, 2), "avg_us": round(us, 2), "FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "WRITE_SIZE_KiB": v.get("WRITE_SIZE"),
                 "hbm_bytes_per_launch": rd + wr, "correction": "read side x2 (gfx950 FETCH_SIZE), write side exact"})
    if per_step >= 0.5: step_bytes += round(per_step) * (rd + wr)
    lines.append("%-60s %6.2f %8.1f %8.1f %8.1f %6.2f" % (k[:60], per_step, us, rd / 1e6, wr / 1e6, (rd + wr) / 1e6 / max(us, 1e-9)))
# the build the counters were taken on (bench.py quotes a table only on the build it was measured with): the id is read from the library file
import re
blob = open("$R/iwae_amd/libiwae_amd.so", "rb").read()
m = re.search(rb"IWAE_BUILD_ID=([0-9a-f]+(?:-diag)?)", blob)
build_id = m.group(1).decode() if m else "unknown"
json.dump({"config": "$CONFIG", "build_id": build_id, "command": "$CMD".replace("$R/", ""), "kernels": kern, "step_hbm_bytes": step_bytes}, open(out + "/kernel_traffic.json", "w"), indent=1)
with open(out + "/kernel_traffic_table.txt", "w") as f:
    f.write("%-60s %6s %8s %8s %8s %6s\n" % ("kernel", "n/step", "avg_us", "rdMB", "wrMB", "TB/s"))
    f.write("\n".join(lines) + "\n")
    f.write("step total (kernels launched every step): %.0f MB\n" % (step_bytes / 1e6))
print(open(out + "/kernel_traffic_table.txt").read())
PY
tail -1 $OUT/trace.log
