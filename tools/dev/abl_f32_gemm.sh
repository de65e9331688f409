#!/bin/bash
# Developer script (GPU box): per-launch times of the float32 step's big GEMMs under the timing ablations of gemm_f32_v2_kernel
# (DIAG build made in the box's copy of the tree; option f32_gemm_dbg: 1 no requests in the k loop, 2 no LDS stores, 4 no MFMAs, 16 no barrier).
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT
cd $R && DIAG=1 bash iwae_amd/csrc/build.sh > gpurun_out/diag_build.log 2>&1 || { tail gpurun_out/diag_build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for d in ${@:-0 1 2 4 16 3 7}; do
  OUT=$R/gpurun_out/abl_f32_$d; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --precision fp32 --steps 12 --warmup 3 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval --opt no_f32_side=1 --opt f32_gemm_dbg=$d > $OUT/log.txt 2>&1 || echo failed
  python3 - "$OUT" $d <<'PY'
import csv, glob, sys, collections
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(fn)):
    if "gemm_f32_v2" in r["Kernel_Name"]:
        key = r["Kernel_Name"].split("gemm_f32_v2_kernel")[1][:22] + " grid " + r.get("Grid_Size_X", "") + "," + r.get("Grid_Size_Y", "") + "," + r.get("Grid_Size_Z", "")
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("dbg", sys.argv[2], "  ".join("%s: %.1f" % (k, sorted(v)[len(v) // 2]) for k, v in sorted(d.items())))
PY
done
