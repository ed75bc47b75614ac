mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04c_t.log 2>&1; echo rc=$? >> gpurun_out/r04c_t.log
tail -4 gpurun_out/r04c_t.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 > gpurun_out/r04c_apdbg.log 2>&1; echo rc=$? >> gpurun_out/r04c_apdbg.log
timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 > gpurun_out/r04c_ap22.log 2>&1; echo rc=$? >> gpurun_out/r04c_ap22.log
PPRHIP_APBS_HOT=0 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 > gpurun_out/r04c_ap22_nohot.log 2>&1; echo rc=$? >> gpurun_out/r04c_ap22_nohot.log
PPRHIP_APBS_HOT=4096 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 > gpurun_out/r04c_ap22_hot4k.log 2>&1; echo rc=$? >> gpurun_out/r04c_ap22_hot4k.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 400 python tools/exp/apbs_rmat24_all.py > gpurun_out/r04c_ap24.log 2>&1; echo rc=$? >> gpurun_out/r04c_ap24.log
timeout -k 10 500 python bench.py --no-cpu-baseline --no-rmat24 --no-pmc --steps 6 --warmup 2 > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; echo rc=$?
