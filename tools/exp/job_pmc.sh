set -o pipefail
root=$(pwd); out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VMEM" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_CYCLE_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES SQ_LEVEL_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  echo "set [$set]"; timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$tag -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > $out/pmc_$tag.log 2>&1 || { echo "set [$set] failed or timed out: stopping"; tail -3 $out/pmc_$tag.log; exit 1; }
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/pmc_$tag/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("pprhip::","")
    if k.startswith(("k_dense_edges_b","k_dense_apply_batch","k_mc_walk")):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k, {c: round(sum(v)/len(v),3) for c,v in acc[k].items()}, flush=True)
PY
  rm -rf $out/pmc_$tag
done
