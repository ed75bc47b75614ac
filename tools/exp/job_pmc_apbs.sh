set -o pipefail
root=$(pwd); out=$root/gpurun_out
export TMPDIR=/tmp PPRHIP_BATCH_THREADS=0
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "VALUBusy SALUBusy MemUnitStalled" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  echo "set [$set]"; timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pmca_$tag -- python3 $root/tools/explore_apbs.py --scale 22 --thr 1e-3 --targets 131072 > $out/pmca_$tag.log 2>&1 || { echo "set [$set] failed or timed out: stopping"; tail -3 $out/pmca_$tag.log; exit 1; }
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/pmca_$tag/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("pprhip::","")
    if k.startswith("k_apbs"):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v)/len(v),3) for c,v in acc[k].items()}, flush=True)
PY
  rm -rf $out/pmca_$tag
done
