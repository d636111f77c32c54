# round 3, call 4L: configs[4] shape on step_main_ws<2> with three-tile rounds (all images in LDS, 90 points per round) against step_main_wp<2>
set -x
mkdir -p gpurun_out/r4l
O=$PWD/gpurun_out/r4l
for w in f32 bf16; do
python bench.py --config stress_256x64 --weights $w --steps 60 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress_wp_$w.json 2>/dev/null; tail -1 $O/bench_stress_wp_$w.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('wp $w', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config stress_256x64 --weights $w --kernel ws1 --steps 60 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress_ws2_$w.json 2>/dev/null; tail -1 $O/bench_stress_ws2_$w.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ws two-tile $w', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config stress_256x64 --weights $w --kernel ws1 --ws-flags 2 --steps 60 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress_ws3_$w.json 2>$O/err_ws3_$w.txt; tail -1 $O/bench_stress_ws3_$w.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ws three-tile $w', j['ms_per_step'], j['roofline']['kernel_ms'])" || tail -3 $O/err_ws3_$w.txt
done
true
