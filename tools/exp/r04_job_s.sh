mkdir -p gpurun_out
: > gpurun_out/r04s.log
for cfg in "1 8192" "2 3072" "2 2048" "1 4096"; do
set -- $cfg
echo "== WGS_PER_CU=$1 HOT=$2" >> gpurun_out/r04s.log
PPRHIP_APBS_WGS_PER_CU=$1 PPRHIP_APBS_HOT=$2 PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "apbs dense\]|searches \+|tier 1|tier 2 \(" | tail -4 | cut -c1-330 >> gpurun_out/r04s.log
done
cat gpurun_out/r04s.log
