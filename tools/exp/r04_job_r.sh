mkdir -p gpurun_out
export PPRHIP_BENCH_WATCHDOG_S=200
timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo > gpurun_out/r04r_gloo2.json 2> gpurun_out/r04r_gloo2.err; echo rc=$?
tail -c 1500 gpurun_out/r04r_gloo2.json
tail -5 gpurun_out/r04r_gloo2.err
