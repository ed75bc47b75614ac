"""Where the device waits in the headline run (developer tool, round 3).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python3 bench.py --steps 2 --warmup 1 ...
    python3 tools/exp/gap_timeline.py /tmp/gt

Merges the kernel intervals of the trace (all streams), lists the time no kernel was running inside the timed region
and groups the idle intervals by the kernel that ended before them and the kernel that started after them.
"""
import collections
import csv
import glob
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pprhip::", "").split("<")[0].strip()


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # the timed region: from the first k_dense_edges_b after the last lift kernel to the end
    lift_end = max((e for s, e, k in rows if k in ("k_build_walk_rec", "k_build_in_rec")), default=rows[0][0])
    rows = [r for r in rows if r[0] >= lift_end]
    t0, t1 = rows[0][0], max(e for s, e, k in rows)
    busy_end, last = rows[0][1], rows[0][2]
    idle = collections.Counter()
    idle_n = collections.Counter()
    total_idle = 0
    for s, e, k in rows[1:]:
        if s > busy_end:
            gap = s - busy_end
            total_idle += gap
            idle[(last, k)] += gap
            idle_n[(last, k)] += 1
        if e > busy_end:
            busy_end, last = e, k
    span = t1 - t0
    # average number of kernels running, and per queue how much of the span it was busy
    ksum = sum(e - s for s, e, k in rows)
    print("kernel time / span = %.2f kernels running on average" % (ksum / span))
    print("span %.1f ms, idle %.1f ms (%.1f %%), %d kernels" % (span / 1e6, total_idle / 1e6, 100.0 * total_idle / span, len(rows)))
    for (a, b), v in idle.most_common(25):
        print("  %-26s -> %-26s %8.2f ms  %6d gaps  avg %7.1f us" % (a, b, v / 1e6, idle_n[(a, b)], v / idle_n[(a, b)] / 1e3))


if __name__ == "__main__":
    main()
