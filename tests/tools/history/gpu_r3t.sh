# round 3, call T: dry run of bench.py's N > 1 path with 2 and 4 ranks on the one GPU of the box (gloo; timings meaningless)
set -x
mkdir -p gpurun_out/r3t
O=$PWD/gpurun_out/r3t
for N in 2 4; do
VMAP_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2957$N bench.py --gpus $N --steps 40 --warmup 5 > $O/bench_gloo_$N.json 2> $O/bench_gloo_$N.err < /dev/null; echo "rc=$?"; tail -1 $O/bench_gloo_$N.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['n_gpus'], j['value'], j['with_background'], j['world'])"; tail -3 $O/bench_gloo_$N.err
done
true
