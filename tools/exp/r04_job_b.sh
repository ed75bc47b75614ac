mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04b_t.log 2>&1; echo rc=$? >> gpurun_out/r04b_t.log
tail -4 gpurun_out/r04b_t.log
timeout -k 10 300 python tools/exp/apbs_big_searches.py 22 > gpurun_out/r04b_big.log 2>&1; echo rc=$? >> gpurun_out/r04b_big.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 > gpurun_out/r04b_apdbg.log 2>&1; echo rc=$? >> gpurun_out/r04b_apdbg.log
timeout -k 10 500 python bench.py --no-cpu-baseline --no-rmat24 --no-pmc --steps 6 --warmup 2 > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; echo rc=$?
