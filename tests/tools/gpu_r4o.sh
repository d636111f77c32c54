# round 3, final tree: rocprofv3 kernel trace + stats of the scored command (bench.py --steps 20 --warmup 5) and of the default command
set -x
mkdir -p gpurun_out/r4o
O=$PWD/gpurun_out/r4o
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_20_5 -o b -- python $R/bench.py --steps 20 --warmup 5 > $O/bench_20_5_under_rocprof.json 2>$O/err1.txt
find $O/prof_20_5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_20_5.csv; head -5 $O/rocprofv3_kernel_stats_20_5.csv | cut -c1-160; rm -rf $O/prof_20_5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_def -o b -- python $R/bench.py > $O/bench_default_under_rocprof.json 2>$O/err2.txt
find $O/prof_def -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_default.csv; head -5 $O/rocprofv3_kernel_stats_default.csv | cut -c1-160; rm -rf $O/prof_def
tail -1 $O/bench_default_under_rocprof.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'])"
true
