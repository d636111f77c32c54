set -x
mkdir -p gpurun_out/ws
export TMPDIR=/tmp
O=$PWD/gpurun_out/ws
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bg -- python $R/bench.py --config background --timed-only --steps 100 --warmup 10 > $O/prof_run.log 2>&1
find $O/prof -name "*kernel_stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 $f
