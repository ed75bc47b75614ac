mkdir -p gpurun_out
bash tools/profile_round.sh r04 apbs > gpurun_out/r04i_prof_apbs.log 2>&1; echo rc=$? >> gpurun_out/r04i_prof_apbs.log
tail -3 gpurun_out/r04i_prof_apbs.log
