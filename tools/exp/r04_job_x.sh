mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "all_pair or apbs or index or multi or shard or rccl or cli" > gpurun_out/r04x_t.log 2>&1; echo rc=$? >> gpurun_out/r04x_t.log; tail -5 gpurun_out/r04x_t.log
: > gpurun_out/r04x.log
for hb in 1 0; do
for t in 262144 524288 1048576 4194304; do
echo "== HELP_BETWEEN=$hb targets $t" >> gpurun_out/r04x.log
PPRHIP_APBS_HELP_BETWEEN=$hb PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank $t 2>&1 | grep -E "apbs dense\]|apbs host\]|\[index\]|targets_per_s" | tail -9 | cut -c1-330 >> gpurun_out/r04x.log
done
done
cat gpurun_out/r04x.log
