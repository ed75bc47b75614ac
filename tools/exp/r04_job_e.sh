mkdir -p gpurun_out
for hot in 8192 4096; do
PPRHIP_APBS_HOT=$hot PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "apbs host|apbs dense\]|index\]|metric" | cut -c1-360 > gpurun_out/r04e_ap22_hot$hot.log
PPRHIP_APBS_HOT=$hot PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "apbs host|apbs dense\]|index\]|targets" | cut -c1-360 > gpurun_out/r04e_ap24_hot$hot.log
done
PPRHIP_APBS_NO_PIPE=1 PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "apbs host|apbs dense\]|index\]|metric" | cut -c1-360 > gpurun_out/r04e_ap22_nopipe.log
