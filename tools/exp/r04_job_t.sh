mkdir -p gpurun_out
: > gpurun_out/r04t.log
for deg in "4,12" "6,12" "8,12" "5,10" "4,16"; do
echo "== DEG=$deg" >> gpurun_out/r04t.log
PPRHIP_APBS_DEG=$deg PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "searches \+|tier 1|tier 2 \(" | tail -3 | cut -c1-200 >> gpurun_out/r04t.log
done
for deg in "4,12" "6,12" "8,16"; do
echo "== R24 DEG=$deg" >> gpurun_out/r04t.log
PPRHIP_APBS_DEG=$deg PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "searches \+|tier 1|tier 2 \(" | tail -3 | cut -c1-200 >> gpurun_out/r04t.log
done
cat gpurun_out/r04t.log
