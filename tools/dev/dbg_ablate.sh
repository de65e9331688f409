#!/bin/bash
# Developer script (GPU box): per-kernel durations under each ablation flag.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for F in 0 1 2 3 4 8 16 32 64 20 127; do
  IWAE_DEBUG_FLAGS=$F timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl/f$F -- python3 $R/tools/dev/dbg_time.py 1024 50 1 ${STEPS:-60} > $R/gpurun_out/abl_f$F.log 2>&1 || echo "flag $F failed"
done
python3 - <<'PY'
import csv, glob, os
R = os.environ['GRAFT_REPO_ROOT']
names = ['out_bwd', 'wgrad', 'dense_kernel<0>', 'dense_kernel<4>', 'dense_kernel<2>', 'dense_kernel<1>', 'dense_kernel<3>', 'sample', 'latent_bwd', 'reduce_grads']
print("flag " + " ".join("%12s" % n[-12:] for n in names))
for F in [0, 1, 2, 3, 4, 8, 16, 32, 64, 20, 127]:
    fs = glob.glob(R + '/gpurun_out/abl/f%d/*/*kernel_stats.csv' % F)
    if not fs: continue
    rows = list(csv.DictReader(open(fs[0])))
    out = []
    for n in names:
        v = [float(r['AverageNs']) / 1e3 for r in rows if n in r['Name']]
        out.append("%12.1f" % v[0] if v else "%12s" % "-")
    print("%4d " % F + " ".join(out))
PY
