mkdir -p gpurun_out
timeout -k 10 400 python tools/exp/two_groups.py 22 4 2 > gpurun_out/r04v_two.log 2>&1; echo rc=$? >> gpurun_out/r04v_two.log
timeout -k 10 400 python tools/exp/two_groups.py 22 4 3 > gpurun_out/r04v_three.log 2>&1; echo rc=$? >> gpurun_out/r04v_three.log
tail -6 gpurun_out/r04v_two.log gpurun_out/r04v_three.log
