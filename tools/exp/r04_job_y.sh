mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04y_t.log 2>&1; echo rc=$? >> gpurun_out/r04y_t.log; tail -5 gpurun_out/r04y_t.log
