#!/bin/bash
# Developer script (GPU box): kernel timeline of one step of bench.py --config c0 (B = 20, k = 1: the reference's default regime).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_c0_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --config c0 --steps 200 --warmup 10 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval $BENCH_ARGS > $OUT/log.txt 2>&1 || echo failed
python3 - "$OUT" <<'PY'
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r["Queue_Id"]) for r in csv.DictReader(open(fn)))
idx = [i for i, e in enumerate(ev) if "wgrad_rows_kernel" in e[2]]
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 9]
t0 = ev[i0][0]; prev = t0
for s, e, n, q in ev[i0:i1]:
    print("%8.1f %7.1f gap %5.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, q, n)); prev = e
PY
