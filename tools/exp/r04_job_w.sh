mkdir -p gpurun_out
: > gpurun_out/r04w.log
for t in 524288 1048576 2097152 4194304; do
echo "== targets $t" >> gpurun_out/r04w.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank $t 2>&1 | grep -E "apbs dense\]|apbs host\]|targets_per_s" | tail -12 | cut -c1-400 >> gpurun_out/r04w.log
done
cat gpurun_out/r04w.log
