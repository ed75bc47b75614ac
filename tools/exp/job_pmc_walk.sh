#!/bin/bash
# Round 3: what bounds k_mc_walk?  Issue counters of the walk kernel in the one-query-at-a-time workload, one rocprofv3
# --pmc pass per set (own runs, no trace domains).   gpurun -- tools/exp/job_pmc_walk.sh
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pw_$tag -- python3 $root/bench.py --mode single --queries-per-step 8 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > /tmp/pw_$tag.log 2>&1 || { echo "set [$set] failed"; tail -3 /tmp/pw_$tag.log; exit 1; }
  python3 - "$tag" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pw_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pprhip::", "")
    if k.startswith(("k_mc_walk", "k_dense_edges<")):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, flush=True)
PY
done
