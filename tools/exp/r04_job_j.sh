mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "all_pair or apbs or rccl or multi or backward or full_size or cli" > gpurun_out/r04j_t.log 2>&1; echo rc=$? >> gpurun_out/r04j_t.log; tail -3 gpurun_out/r04j_t.log
for deg in "4,12" "3,9" "6,16" "4,24" "0,0"; do
PPRHIP_APBS_DEG=$deg PPRHIP_APBS_NO_PIPE=1 PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "searches \+|metric|tier 1|tier 2" | tail -4 | cut -c1-220 > gpurun_out/r04j_ap22_nopipe_$deg.log
done
PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "part [0-9]|part by part|searches \+|metric" | tail -6 | cut -c1-220 > gpurun_out/r04j_ap22_pipe.log
PPRHIP_APBS_NO_PIPE=1 PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "searches \+|targets|tier 1|tier 2" | tail -4| cut -c1-220 > gpurun_out/r04j_ap24_nopipe.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "part [0-9]|part by part|searches \+|targets" | tail -6 | cut -c1-220 > gpurun_out/r04j_ap24_pipe.log
