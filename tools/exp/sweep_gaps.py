"""What the other streams did while the compute stream waited (developer tool, round 5).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python3 bench.py --trace-child ...
    python3 tools/exp/sweep_gaps.py /tmp/gt [pairs to show, default 3 per kind]

The compute stream is the one the batched sweeps (k_dense_edges_b) run on.  For the kinds of gap that cost most (kernel
before -> kernel after) the script prints a few examples: every kernel of any stream that overlaps the window around
the gap, with times relative to the gap's start - i.e. what the host had queued elsewhere while it did not feed the
sweeps' stream.
"""
import collections
import csv
import glob
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pprhip::", "").split("<")[0].strip()


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    main_stream = collections.Counter(s for a, b, k, s in rows if k == "k_dense_edges_b").most_common(1)[0][0]
    first = min(a for a, b, k, s in rows if k == "k_dense_edges_b")
    rows = [r for r in rows if r[0] >= first]
    on_main = [r for r in rows if r[3] == main_stream]
    gaps = collections.defaultdict(list)
    for p, q in zip(on_main, on_main[1:]):
        if q[0] > p[1]:
            gaps[(p[2], q[2])].append((q[0] - p[1], p[1], q[0]))
    kinds = sorted(gaps.items(), key=lambda kv: -sum(g[0] for g in kv[1]))[:6]
    for (a, b), gl in kinds:
        print("== %s -> %s: %d gaps, %.2f ms" % (a, b, len(gl), sum(g[0] for g in gl) / 1e6))
        for gap, t0, t1 in sorted(gl, reverse=True)[:show]:
            print("   gap %.1f us" % (gap / 1e3))
            for s, e, k, st in rows:
                if e >= t0 - 60000 and s <= t1 + 20000:
                    print("      %9.1f .. %9.1f us  stream %-3s %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, st, k, "   <== compute stream" if st == main_stream else ""))


if __name__ == "__main__":
    main()
