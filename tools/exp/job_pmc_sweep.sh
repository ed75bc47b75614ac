#!/bin/bash
# Issue and memory-path counters of the batched sweep's kernels (k_dense_edges_b / k_dense_apply_batch) in the headline
# workload, one rocprofv3 --pmc pass per counter set (own runs, no trace domains).   gpurun -- tools/exp/job_pmc_sweep.sh
set -o pipefail
export PPRHIP_LIB_PATH=${PPRHIP_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/personalized-pagerank-algorithms-on-neo4j_amd/libpprhip_hooks.so}  # (the switches these jobs set are test hooks)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
for parts in 0; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  tag=p${parts}_$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/ps_$tag -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > /tmp/ps_$tag.log 2>&1 || { echo "parts=$parts set [$set] failed"; tail -3 /tmp/ps_$tag.log; continue; }
  python3 - "$tag" "$parts" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/ps_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pprhip::", "")
    if k.startswith(("k_dense_edges_b", "k_dense_apply_batch")):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print("parts=%s" % sys.argv[2], k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, flush=True)
PY
done
done
