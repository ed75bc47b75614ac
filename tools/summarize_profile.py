"""Condenses the rocprofv3 output of tools/profile_round.sh into profiles/ (developer tool).

    python tools/summarize_profile.py <tag> <workload> [<workload> ...]      e.g.  r02 bench single topk apbs

For every workload it reads gpurun_out/<tag>_<workload>_{stats,fetch,write}/ and writes
  profiles/<tag>_<workload>_kernel_stats.csv   the `rocprofv3 --kernel-trace --stats` summary, verbatim;
  profiles/<tag>_<workload>_summary.md         the same per kernel with the FETCH_SIZE / WRITE_SIZE averages of the two
                                               `--pmc` passes and what follows from them.

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KB and are
collected in separate passes (they do not fit one pass on gfx950); FETCH_SIZE = TCC_EA0_RDREQ x 64 B, while every
memory-side read request of gfx950 is 128 bytes, coalesced stream and random gather alike
(profiles/r02_fetch_calibration.txt; re-checked here on k_sum_partial, which streams exactly 8n bytes).  So
FETCH_SIZE x 1024 / 64 is the number of 128-byte lines that left L2 and read bytes = 2 x FETCH_SIZE; about 52-55 G
random lines per second is what the chip delivers (tools/micro/gather_rate.hip, row_gather_rate.hip) = 6.7-7 TB/s.
"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMANDS = {
    "bench": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras",
    "single": "python3 bench.py --mode single --queries-per-step 16 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc "
              "--no-extras",
    "topk": "python3 tools/bench_topk.py 22 64",
    "apbs": "python3 tools/explore_apbs.py --scale 22 --thr 1e-3 --targets 262144",
}


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pprhip::", "").strip()


def pmc(d, counter):
    out = collections.defaultdict(list)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return out
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def summarize(tag, what):
    base = os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, what))
    sfiles = glob.glob(os.path.join(base + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if not sfiles:
        print("no stats for", what)
        return
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    shutil.copy(sfiles[0], os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, what)))
    stats = list(csv.DictReader(open(sfiles[0])))
    fetch = pmc(base + "_fetch", "FETCH_SIZE")
    write = pmc(base + "_write", "WRITE_SIZE")
    n = 1 << 22
    lines = ["# rocprofv3 summary `%s_%s`" % (tag, what), "",
             "Command: `rocprofv3 --kernel-trace --stats -- %s`; PMC passes `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, "
             "each its own run (tools/profile_round.sh).  R-MAT scale 22 (n = 4 194 304, m = 67 108 864)."
             % COMMANDS.get(what, what), "",
             "| kernel | calls | total ms | avg us | % | FETCH_SIZE avg KB | WRITE_SIZE avg KB | 128-B lines beyond L2 per s | HBM-side GB/s (2 x FETCH + WRITE) |",
             "|---|---|---|---|---|---|---|---|---|"]
    for r in stats:
        k = short(r["Name"])
        f, w = fetch.get(k, []), write.get(k, [])
        avg_us = float(r["AverageNs"]) / 1e3
        fk = sum(f) / len(f) if f else None
        req = "%.1f G" % (fk * 1024.0 / 64.0 / (avg_us * 1e-6) / 1e9) if fk and avg_us > 0 else "-"
        wk = sum(w) / len(w) if w else None
        gbs = "%.0f" % ((2.0 * (fk or 0.0) + (wk or 0.0)) * 1024.0 / (avg_us * 1e-6) / 1e9) if (fk or wk) and avg_us > 0 else "-"
        lines.append("| %s | %s | %.3f | %.1f | %s | %s | %s | %s | %s |" % (
            k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, avg_us, r["Percentage"],
            "%.0f" % fk if fk is not None else "-", "%.0f" % wk if wk is not None else "-", req, gbs))
    cal = fetch.get("k_sum_partial", [])
    if cal:
        # sums inside the queries stop at the last non-isolated node; only a launch over all n values calibrates
        ratio = max(cal) * 1024.0 / (8.0 * n)
        if ratio > 0.45:
            lines += ["", "Calibration: `k_sum_partial` over a whole vector streams exactly 8n = %d bytes; FETCH_SIZE "
                      "reports %.0f KB = %.3f of it (128-byte requests tallied at 64: the guide's gfx950 correction)."
                      % (8 * n, max(cal), ratio)]
        else:
            lines += ["", "(`k_sum_partial` only runs over the live range of ids in this workload: %.0f KB of FETCH_SIZE "
                      "per launch for 8 x n_live bytes; the calibration launch over all n values is in the bench "
                      "workload's summary.)" % max(cal)]
    open(os.path.join(ROOT, "profiles", "%s_%s_summary.md" % (tag, what)), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


def main():
    tag = sys.argv[1]
    for what in sys.argv[2:]:
        summarize(tag, what)


if __name__ == "__main__":
    main()
