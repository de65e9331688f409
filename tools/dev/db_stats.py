"""Developer script: per-kernel summary of a rocprofv3 `*_results.db` (the default output format of `rocprofv3 --kernel-trace`):
  python tools/dev/db_stats.py path/to/x_results.db [tail_fraction] [--seq N]
tail_fraction (default 0.5): only the dispatches of the last part of the run are counted (skips warm-up).  --seq N prints the last N dispatches
in start order with their offsets (a timeline of one step)."""
import collections
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 0.5
    seq = int(sys.argv[sys.argv.index("--seq") + 1]) if "--seq" in sys.argv else 0
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [x for x in names if x.startswith("rocpd_kernel_dispatch")][0]
    ks = [x for x in names if x.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute("select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.queue_id "
                          "from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
    tail = rows[int(len(rows) * (1.0 - frac)):]
    agg = collections.OrderedDict()
    for r in tail:
        key = re.sub(r"\(.*", "", r[0])[:72]
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (r[2] - r[1]) / 1e3
    span = (tail[-1][2] - tail[0][1]) / 1e3
    busy = sum(v[1] for v in agg.values())
    print("%d dispatches, span %.1f us, sum of kernel durations %.1f us (%.2f x)" % (len(tail), span, busy, busy / span))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-74s n=%5d avg=%8.1f us  total=%10.0f us  %5.1f %%" % (k, v[0], v[1] / v[0], v[1], 100.0 * v[1] / busy))
    if seq:
        t0 = rows[-seq][1]
        for r in rows[-seq:]:
            print("%9.1f %8.1f  q%-2d %-60s grid=(%d,%d,%d)/%d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[7], re.sub(r"\(.*", "", r[0])[:60],
                                                                  r[3], r[4], r[5], r[6]))


if __name__ == "__main__":
    main()
