# round 4, last session: waves per CU of the walk kernel in top-k rounds (short phases: 6 walks per lane at 8 waves per CU)
mkdir -p gpurun_out
for w in 8 4 3 2 1 16; do
  echo "== PPRHIP_TOPK_WALK_WAVES=$w" >> gpurun_out/s3_walkwaves.log
  PPRHIP_TOPK_WALK_WAVES=$w timeout -k 10 200 python tools/bench_topk.py 22 128 2>/dev/null | grep "single\|walk phases" >> gpurun_out/s3_walkwaves.log
done
cat gpurun_out/s3_walkwaves.log
