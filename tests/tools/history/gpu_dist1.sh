# the N-GPU code path of bench.py with one rank (RCCL initialised, collectives are identities)
set -x
mkdir -p gpurun_out/dist1
export TMPDIR=/tmp
O=$PWD/gpurun_out/dist1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline > $O/bench_torchrun.json 2> $O/bench_torchrun.err < /dev/null; tail -1 $O/bench_torchrun.json | head -c 300; echo
VMAP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_bg_dist1.json 2> $O/bench_bg_dist1.err < /dev/null
tail -1 $O/bench_bg_dist1.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['with_background']['ms_per_step'], j['world'])"
tail -3 $O/bench_torchrun.err
