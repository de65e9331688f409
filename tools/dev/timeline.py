"""Developer script: print the kernel timeline of one train step from a rocprofv3 --kernel-trace CSV.
usage: python tools/dev/timeline.py <dir containing *kernel_trace.csv> [step index]"""
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:58], r["Queue_Id"]) for r in csv.DictReader(open(fn)))
starts = [i for i, e in enumerate(ev) if "prep_rows" in e[2] or "gather_binarize" in e[2]]
if len(starts) < 4: starts = [i for i, e in enumerate(ev) if "block_fwd_kernel" in e[2]]      # the fused encoder kernel converts the batch itself
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
i0, i1 = starts[n], starts[n + 1]
t0 = ev[i0][0]
busy, cur = 0, t0
for s, e, name, q in ev[i0:i1]:
    gap = max(0, s - cur)
    print("%8.1f %7.1f  gap %5.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, name))
    if e > cur:
        busy += e - max(s, cur)
        cur = e
print("period %.1f us, busy (union) %.1f us" % ((ev[i1][0] - t0) / 1e3, busy / 1e3))
