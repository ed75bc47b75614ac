"""Which kernels run while a walk kernel runs? (developer tool, round 3: the next top-k round's push beside the walks)

    rocprofv3 --kernel-trace --output-format csv -d /tmp/wo -- python3 tools/prof_topk_single.py
    python3 tools/exp/walk_overlap.py /tmp/wo
"""
import collections
import csv
import glob
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pprhip::", "").split("<")[0].strip()


f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
walks = [r for r in rows if r[2] == "k_mc_walk"]
inside = collections.Counter()
inside_ns = collections.Counter()
tot = 0
for ws, we, _, wq in walks:
    tot += we - ws
    for s, e, k, q in rows:
        if k != "k_mc_walk" and s < we and e > ws:
            inside[(k, q == wq)] += 1
            inside_ns[(k, q == wq)] += min(e, we) - max(s, ws)
print("%d walk kernels, %.1f ms in total, avg %.0f us; queues seen: %s" % (len(walks), tot / 1e6, tot / 1e3 / max(1, len(walks)), sorted(set(r[3] for r in rows))))
for (k, same), c in inside.most_common(12):
    print("  %-24s %-11s %5d launches, %8.2f ms inside walk kernels" % (k, "same queue" if same else "other queue", c, inside_ns[(k, same)] / 1e6))
# a sample round: the kernels around the 10th walk kernel
if len(walks) > 10:
    ws, we = walks[10][0], walks[10][1]
    print("around walk #10 (t = 0 at its start, us):")
    for s, e, k, q in rows:
        if s > ws - 400000 and s < we + 300000:
            print("   %8.1f .. %8.1f  q%-3s %s" % ((s - ws) / 1e3, (e - ws) / 1e3, q, k))

# per walk kernel: its duration, the span of the other queue's kernels that started while it ran or before the next
# walk kernel, and the time to the next walk kernel
print("per round: walk us | other-queue kernels: first start, last end (us after the walk's start), busy us | next walk starts")
for i in range(8, min(len(walks) - 1, 40)):
    ws, we, _, wq = walks[i]
    nxt = walks[i + 1][0]
    other = [(s, e) for s, e, k, q in rows if q != wq and s >= ws and s < nxt]
    same = [(s, e, k) for s, e, k, q in rows if q == wq and s >= we and s < nxt]
    ob = sum(e - s for s, e in other)
    sb = sum(e - s for s, e, k in same)
    print("  walk %6.0f | other %6.0f .. %6.0f busy %6.0f | same-queue busy after walk %6.0f | next walk at %6.0f"
          % ((we - ws) / 1e3, (other[0][0] - ws) / 1e3 if other else -1, (other[-1][1] - ws) / 1e3 if other else -1, ob / 1e3, sb / 1e3, (nxt - ws) / 1e3))
