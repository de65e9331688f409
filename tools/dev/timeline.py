"""Developer script: print the kernel timeline of one train step from a rocprofv3 --kernel-trace CSV.
usage: python tools/dev/timeline.py <dir containing *kernel_trace.csv> [step index]"""
import csv, glob, sys
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:58], r["Queue_Id"]) for r in csv.DictReader(open(fn)))
# one step = from one reduce on the main queue to the next (exactly one per training step; the window is a whole step, phase-shifted:
# it opens with the encoder forward of the next step)
anchor = "latent_bwd_kernel" if any("latent_bwd_kernel" in e[2] for e in ev) else "block_bwd_kernel"      # (round 5: the latent sums may ride in block_bwd_kernel<4>)
starts = [i + 1 for i, e in enumerate(ev) if anchor in e[2]]
starts = [i for i in starts if i < len(ev)]
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
