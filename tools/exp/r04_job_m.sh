mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r04m_t.log 2>&1; echo rc=$? >> gpurun_out/r04m_t.log; tail -3 gpurun_out/r04m_t.log
timeout -k 10 900 python bench.py > gpurun_out/r04m_bench.json 2> gpurun_out/r04m_bench.err; echo bench rc=$?
tail -3 gpurun_out/r04m_bench.err
