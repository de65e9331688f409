#!/bin/bash
# Developer script (GPU box): duration of kernels ALONE on the machine (a --pmc run serialises the dispatches) for a list of
# option variants (iwae_set_option switches).  usage: tools/dev/alone.sh "<kernel substrings, | separated>" "name=1" "a=2 b=3" ...
PATS=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/alone_tmp
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf $OUT; mkdir -p $OUT
  opts=""; for o in $1; do opts="$opts --opt $o"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $R/bench.py --steps 30 --warmup 10 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval $opts $BENCH_ARGS > $OUT/log.txt 2>&1 || echo "run failed: $1"
  python3 - "$OUT" "$PATS" "$1" <<'PY'
import csv, glob, sys, collections
out, pats, tag = sys.argv[1], sys.argv[2].split("|"), sys.argv[3]
dur = collections.defaultdict(list)
for fn in glob.glob(out + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = []
for k, d in dur.items():
    if any(p in k for p in pats):
        d = sorted(d); res.append("%s=%.1f" % (k.replace("void iwae::", "").split("(")[0][:40], d[len(d) // 2]))
print("%-40s %s" % (tag, "  ".join(sorted(res))))
PY
}
run ""
for v in "$@"; do run "$v"; done
