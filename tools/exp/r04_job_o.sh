mkdir -p gpurun_out
for w in bench single topk apbs; do
  bash tools/profile_round.sh r04 $w > gpurun_out/r04o_prof_$w.log 2>&1; echo "$w rc=$?" >> gpurun_out/r04o_prof_$w.log
  tail -2 gpurun_out/r04o_prof_$w.log
done
