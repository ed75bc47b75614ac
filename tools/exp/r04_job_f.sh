mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_scale.py -m gpu -x -q -k "all_pair or apbs or rccl or multi or backward" > gpurun_out/r04f_t.log 2>&1; echo rc=$? >> gpurun_out/r04f_t.log; tail -3 gpurun_out/r04f_t.log
for parts in "0.125,0.56" "0.333,0.667" "0.08,0.54" "0.2,0.6"; do
PPRHIP_APBS_PARTS=$parts PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "part [0-9]|part by part|searches \+|index\]|metric" | cut -c1-220 > gpurun_out/r04f_ap22_$parts.log
done
for parts in "0.125,0.56" "0.333,0.667"; do
PPRHIP_APBS_PARTS=$parts PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "part [0-9]|part by part|searches \+|index\]|targets" | cut -c1-220 > gpurun_out/r04f_ap24_$parts.log
done
timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | tail -1 | cut -c1-250 > gpurun_out/r04f_ap22_clean.log
timeout -k 10 400 python tools/exp/apbs_rmat24_all.py 2>&1 | tail -1 > gpurun_out/r04f_ap24_clean.log
