# wave-cycle accounting of every kernel of a workload: active / waiting-for-issue / waiting-for-anything shares
set -o pipefail
root=$(pwd); out=$root/gpurun_out
export TMPDIR=/tmp PPRHIP_BATCH_THREADS=0
cd /tmp
what=$1; shift
prog=$root/$1; shift
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmcw_$what -- python3 $prog "$@" > $out/pmcw_$what.log 2>&1 || { echo "failed"; tail -3 $out/pmcw_$what.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/pmcw_$what/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("pprhip::","")
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
rows=sorted(acc.items(), key=lambda kv:-kv[1]["SQ_WAVE_CYCLES"])
print("%-34s %8s %14s %7s %9s %9s" % ("kernel","launches","wave_cycles","active","wait_inst","wait_any"))
for k,c in rows[:16]:
    w=c["SQ_WAVE_CYCLES"] or 1.0
    print("%-34s %8d %14.3e %7.3f %9.3f %9.3f" % (k[:34], n[k], w, c["SQ_ACTIVE_INST_ANY"]/w, c["SQ_WAIT_INST_ANY"]/w, c["SQ_WAIT_ANY"]/w), flush=True)
PY
rm -rf $out/pmcw_$what
