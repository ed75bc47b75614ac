mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "all_pair or apbs or rccl or multi or backward or full_size or cli" > gpurun_out/r04h_t.log 2>&1; echo rc=$? >> gpurun_out/r04h_t.log; tail -3 gpurun_out/r04h_t.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "part [0-9]|part by part|searches \+|index\]|metric|tier 1" | cut -c1-220 > gpurun_out/r04h_ap22.log
PPRHIP_APBS_NO_PIPE=1 PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "part [0-9]|part by part|searches \+|index\]|metric|tier 1|tier 2" | cut -c1-220 > gpurun_out/r04h_ap22_nopipe.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "part [0-9]|part by part|searches \+|index\]|targets|tier 1" | cut -c1-220 > gpurun_out/r04h_ap24.log
timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | tail -1 | cut -c1-250 > gpurun_out/r04h_ap22_clean.log
timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | tail -1 > gpurun_out/r04h_ap24_clean.log
