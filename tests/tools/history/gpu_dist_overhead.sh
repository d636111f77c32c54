mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
python bench.py --steps 800 --warmup 80 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys;d=json.loads(sys.stdin.read());print('plain', round(d['value']/1e6,2), d['ms_per_step'])"
VMAP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 python bench.py --steps 800 --warmup 80 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys;d=json.loads(sys.stdin.read());print('dist1', round(d['value']/1e6,2), d['ms_per_step'])"
done
