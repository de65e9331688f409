#!/bin/bash
# Profiles the default bench command on the GPU box and writes summaries under gpurun_out/profile_<tag>/
# (copy the ones to keep into profiles/).  Kernel-trace/stats and PMC runs are separate invocations.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-llh-eval"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace run failed"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_mfma -- $CMD > $OUT/pmc_mfma.log 2>&1 || echo "pmc mfma failed"
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
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
traffic = {}
for short, pat in (("bernoulli_fwd", "bern_pipe_kernel<7, true"), ("out_bwd", "out_bwd"), ("wgrad_out", "wgradp_kernel<16, true")):
    ks = [k for k in summ if pat in k]
    if not ks: continue
    v = summ[ks[0]]
    hbm = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
    traffic[short] = {"kernel": ks[0], "FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "WRITE_SIZE_KiB": v.get("WRITE_SIZE"),
                      "hbm_bytes_per_launch": hbm, "correction": "read side x2 (gfx950 FETCH_SIZE), write side exact"}
    print("%s HBM bytes/launch: %.1f MB (fetch %.1f KiB raw, write %.1f KiB)" % (short, hbm / 1e6, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
json.dump(traffic, open(out + "/kernel_traffic.json", "w"), indent=1)
for r in rows[:14]:
    print("%-62s calls %5s avg_us %9.2f pct %6s" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 $OUT/trace.log
