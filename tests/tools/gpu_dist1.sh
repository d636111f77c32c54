mkdir -p gpurun_out; export TMPDIR=/tmp
VMAP_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/bench_dist1.log | cut -c1-400
