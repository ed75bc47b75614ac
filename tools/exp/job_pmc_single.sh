#!/bin/bash
# which counter saturates in the single-query sweep?  Issue and memory-path counters of
# k_dense_edges<true, true> / k_dense_apply in one-query-at-a-time FORA (bench.py --mode single), one rocprofv3 --pmc
# pass per counter set (own runs, no trace domains).   gpurun -- tools/exp/job_pmc_single.sh  -> gpurun_out/r06_pmc_single.txt
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/${1:-r06_pmc_single}.txt
export TMPDIR=/tmp
cd /tmp
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  tag=s_$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/ps_$tag -- python3 $root/bench.py --mode single --steps 1 --warmup 1 --queries-per-step 8 --no-cpu-baseline --no-pmc --no-extras --no-rmat24 > /tmp/ps_$tag.log 2>&1 || { echo "set [$set] failed: $(tail -2 /tmp/ps_$tag.log | tr '\n' ' ')" >> $out; continue; }
  python3 - "$tag" >> $out <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/ps_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pprhip::", "")
    if k.startswith(("k_dense_edges<", "k_dense_edges_panel", "k_panel_fold", "k_dense_apply<", "k_dense_reduce")):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, "launches", len(next(iter(acc[k].values()))), flush=True)
PY
done
cat $out
