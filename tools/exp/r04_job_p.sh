mkdir -p gpurun_out
cat /sys/kernel/mm/transparent_hugepage/enabled > gpurun_out/r04p_thp.txt 2>&1
for i in 1 2; do
PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "searches \+|metric|tier 1|tier 2|index\]" | tail -7 | cut -c1-220 >> gpurun_out/r04p_ap22.log
done
PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "searches \+|targets|tier 1|tier 2|index\]" | tail -7| cut -c1-220 > gpurun_out/r04p_ap24.log
