#!/bin/bash
# Developer script (GPU box): kernel timeline of one float32 training step (bench.py --precision fp32), with grid sizes.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_f32_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --precision fp32 --steps 20 --warmup 5 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval $BENCH_ARGS > $OUT/log.txt 2>&1 || echo failed
python3 - "$OUT" <<'PY'
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r["Queue_Id"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))) for r in rows)
idx = [i for i, e in enumerate(ev) if "lse_kernel" in e[2]]
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = ev[i0][0]; prev = t0
for s, e, n, q, gx, gy, gz, wg in ev[i0:i1]:
    print("%8.1f %7.1f gap %5.1f q%s grid %s,%s,%s wg %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, q, gx, gy, gz, wg, n)); prev = max(prev, e)
print("period %.1f us" % ((ev[i1][0] - t0) / 1e3))
PY
