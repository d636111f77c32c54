# round 3, call 4I: the reference's real iMAP batch (1 x 4800 rays x 14, hidden 256): exact-fp32 kernels vs the eight-wave step_main_ws with several rounds per workgroup
set -x
mkdir -p gpurun_out/r4i
O=$PWD/gpurun_out/r4i
for k in auto gen wide ws1; do
python bench.py --config imap_full --kernel $k --steps 60 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $O/bench_imap_full_$k.json 2>$O/err_$k.txt; tail -1 $O/bench_imap_full_$k.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$k', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel'][:50])" || tail -3 $O/err_$k.txt
done
true
