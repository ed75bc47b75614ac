"""Condenses rocprofv3 output of `bench.py` into profiles/ (developer tool).

    python tools/summarize_profile.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [bench.json]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_summary.md (per-kernel table incl. PMC averages) and updates profiles/traffic.json
(per-launch HBM bytes of the dominant kernels, read by bench.py for roofline.traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KB,
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts coalesced streaming reads at half
their size (checked here on k_sum_partial, which reads exactly 8n bytes) while 64-byte random
gathers are counted in full, so corrected = FETCH_SIZE*1024 + streaming_bytes/2 + WRITE_SIZE*1024
with streaming_bytes taken from the kernel's known sequential reads.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pprhip::", "").strip()


def pmc(d, counter):
    out = collections.defaultdict(list)
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        return out
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    bench_json = sys.argv[5] if len(sys.argv) > 5 else None
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    sfile = glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv"))[0]
    shutil.copy(sfile, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    stats = list(csv.DictReader(open(sfile)))
    fetch = pmc(fetch_dir, "FETCH_SIZE")
    write = pmc(write_dir, "WRITE_SIZE")
    scale, n, m = 22, 1 << 22, 16 << 22
    if bench_json and os.path.exists(bench_json):
        b = json.load(open(bench_json))
        wl = b["config"]["workload"]
        scale = int(wl.split("scale-")[1].split(" ")[0])
        n, m = 1 << scale, 16 << scale
    lines = ["# rocprofv3 summary `%s`" % tag, "",
             "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` "
             "(PMC passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each its own run with --kernel-trace only).", "",
             "| kernel | calls | total ms | avg us | % | FETCH_SIZE avg KB | WRITE_SIZE avg KB |", "|---|---|---|---|---|---|---|"]
    for r in stats:
        k = short(r["Name"])
        f = fetch.get(k, [])
        w = write.get(k, [])
        lines.append("| %s | %s | %.3f | %.1f | %s | %s | %s |" % (
            k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"],
            "%.0f" % (sum(f) / len(f)) if f else "-", "%.0f" % (sum(w) / len(w)) if w else "-"))
    # calibration + traffic of the dense level
    cal = fetch.get("k_sum_partial", [])
    lines += ["", "## HBM traffic"]
    if cal:
        ratio = (sum(cal) / len(cal)) * 1024.0 / (8.0 * n)
        lines.append("Calibration: `k_sum_partial` streams exactly 8n = %d bytes and FETCH_SIZE reports %.0f KB = "
                     "%.3f of it, i.e. the guide's gfx950 half-count holds for this engine's coalesced reads."
                     % (8 * n, sum(cal) / len(cal), ratio))
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))

    def avg(d, k):
        v = d.get(k, [])
        return sum(v) / len(v) if v else 0.0

    dense_keys = [k for k in fetch if k.startswith("k_dense_tiles") or k.startswith("k_dense_edges<")
                  or k.startswith("k_dense_apply<") or k == "k_dense_reduce"]
    if any(k.startswith("k_dense_edges<") or k.startswith("k_dense_tiles") for k in dense_keys):
        raw_f = sum(avg(fetch, k) for k in dense_keys) * 1024.0
        raw_w = sum(avg(write, k) for k in dense_keys) * 1024.0
        # in-edge columns, row-start flag bits, and per row: row id, row sum, residue, reserve, packed out extent
        streaming = 4.0 * m + m / 8.0 + (4.0 + 8.0 + 8.0 + 8.0 + 8.0) * n
        corrected = raw_f + streaming / 2.0 + raw_w
        alg = 12 * m + 36 * n + 4
        lines.append("Dense pull level (%s): FETCH_SIZE %.0f MB + WRITE_SIZE %.0f MB raw; sequential reads of the "
                     "level = %.0f MB, half of which FETCH_SIZE misses => corrected HBM traffic %.0f MB per level vs "
                     "%.0f MB algorithmic (x%.2f)." % (" + ".join(sorted(dense_keys)), raw_f / 1e6, raw_w / 1e6,
                                                       streaming / 1e6, corrected / 1e6, alg / 1e6, corrected / alg))
        traffic.setdefault("dense_pull", {})["scale%d" % scale] = int(corrected)
    batch_keys = [k for k in fetch if k.startswith("k_dense_edges_b<") or k in ("k_dense_apply_batch",
                                                                                "k_dense_reduce_batch")]
    if batch_keys:
        raw_f = sum(avg(fetch, k) for k in batch_keys) * 1024.0
        raw_w = sum(avg(write, k) for k in batch_keys) * 1024.0
        # coalesced reads of a sweep: in-edge columns + flag bits (edge kernel); row sums, row ids, degrees and
        # the slots' residue / reserve vectors (apply kernel, upper bound: every slot busy and crossing)
        B = 16
        streaming = 4.0 * m + m / 8.0 + (8.0 * B + 4.0 + 8.0 + 16.0 * B) * n
        corrected = raw_f + streaming / 2.0 + raw_w
        lines.append("Batched dense sweep (%s): FETCH_SIZE %.0f MB + WRITE_SIZE %.0f MB raw per sweep; coalesced "
                     "reads of a sweep <= %.0f MB, half of which FETCH_SIZE misses => corrected HBM-side traffic "
                     "<= %.0f MB per sweep (the gathers' 128-byte lines are counted in full)."
                     % (" + ".join(sorted(batch_keys)), raw_f / 1e6, raw_w / 1e6, streaming / 1e6, corrected / 1e6))
        traffic.setdefault("dense_pull_batch", {})["scale%d" % scale] = int(corrected)
    if "k_mc_walk" in fetch:
        raw = avg(fetch, "k_mc_walk") * 1024.0 + avg(write, "k_mc_walk") * 1024.0
        lines.append("Walk kernel: FETCH_SIZE + WRITE_SIZE = %.0f MB per launch (random 64-byte requests, counted "
                     "in full)." % (raw / 1e6))
        traffic.setdefault("walk", {})["scale%d" % scale] = int(raw)
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    if bench_json and os.path.exists(bench_json):
        lines += ["", "## bench.py line of the same build", "```json", open(bench_json).read().strip(), "```"]
        shutil.copy(bench_json, os.path.join(ROOT, "profiles", tag + "_bench.json"))
    open(os.path.join(ROOT, "profiles", tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))


if __name__ == "__main__":
    main()
