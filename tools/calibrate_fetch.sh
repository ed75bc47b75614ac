#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE on gfx950 for random row gathers of known size (tools/micro/row_gather_rate.hip:
# every launch gathers exactly 1 073 741 824 bytes as rows of 8 ... 512 bytes from a table far beyond L2), next to the
# L2's memory-side read-request counters by request size.  Run from the repo root through gpurun; writes
# gpurun_out/fetch_calibration.txt (copied to profiles/ by hand).
set -o pipefail
root=$(pwd); out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- $root/tools/micro/row_gather_rate > $out/cal_fetch.log 2>&1 || { echo "FETCH_SIZE pass failed"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/cal_rdreq -- $root/tools/micro/row_gather_rate > $out/cal_rdreq.log 2>&1 || { echo "RDREQ pass failed"; exit 1; }
python3 - <<PY > $out/fetch_calibration.txt
import csv, glob, collections
def load(d):
    f = glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, rd = load("cal_fetch"), load("cal_rdreq")
true_bytes = 2048 * 256 * 32 * 8 * 8
print("random row gathers, %d bytes gathered per launch (6 launches per row width: 537 MB and 2147 MB tables, averaged)" % true_bytes)
print("row_B  FETCH_SIZE_KB  FETCH_SIZE/true  RDREQ  RDREQ_128B  RDREQ_64B  RDREQ_32B  (128*n128+64*n64+32*n32)/true")
for g in (1, 2, 4, 8, 16, 32, 64):
    k = "k_rows<%d>" % g
    f = sum(fe[k]["FETCH_SIZE"]) / len(fe[k]["FETCH_SIZE"])
    c = {n: sum(v) / len(v) for n, v in rd[k].items()}
    by = 128 * c["TCC_EA0_RDREQ_128B_sum"] + 64 * c["TCC_EA0_RDREQ_64B_sum"] + 32 * c["TCC_EA0_RDREQ_32B_sum"]
    print("%5d  %13.0f  %15.3f  %.3e  %.3e  %.3e  %.3e  %.3f" % (8 * g, f, f * 1024 / true_bytes, c["TCC_EA0_RDREQ_sum"],
          c["TCC_EA0_RDREQ_128B_sum"], c["TCC_EA0_RDREQ_64B_sum"], c["TCC_EA0_RDREQ_32B_sum"], by / true_bytes))
PY
cat $out/fetch_calibration.txt
rm -rf $out/cal_fetch $out/cal_rdreq
