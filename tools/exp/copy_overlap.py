"""Do the delivery copies overlap the compute kernels? (developer tool, round 3)

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/co -- python3 tools/exp/fetch_rate.py 22 64
    python3 tools/exp/copy_overlap.py /tmp/co

For every device-to-host copy of more than 1 MB: its duration and the share of it during which some kernel ran.
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    cf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0]
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kf)))
    # merged busy intervals
    busy = []
    for s, e in ks:
        if busy and s <= busy[-1][1]:
            busy[-1][1] = max(busy[-1][1], e)
        else:
            busy.append([s, e])
    rows = list(csv.DictReader(open(cf)))
    print("copy trace columns:", list(rows[0].keys()))
    big = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s > 200000:
            big.append((s, e, r.get("Direction", "?")))
    import bisect
    starts = [b[0] for b in busy]
    tot = cov = 0
    for s, e, _ in big:
        i = max(0, bisect.bisect_right(starts, s) - 1)
        c = 0
        while i < len(busy) and busy[i][0] < e:
            c += max(0, min(e, busy[i][1]) - max(s, busy[i][0]))
            i += 1
        tot += e - s
        cov += c
    print("%d long copies, %.1f ms in total (avg %.0f us), kernels ran during %.1f %% of that time"
          % (len(big), tot / 1e6, tot / 1e3 / max(1, len(big)), 100.0 * cov / max(1, tot)))
    dirs = {}
    for s, e, dr in big:
        dirs[dr] = dirs.get(dr, 0) + 1
    print(dirs)


if __name__ == "__main__":
    main()
