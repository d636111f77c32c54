# rocprofv3 per-kernel summary of the headline bench command on the end-of-round library
set -x
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=$PWD/gpurun_out/final
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o hl -- python $R/bench.py --timed-only > $O/prof_run.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
cd $R
for f in $O/prof/*kernel_stats.csv $O/prof205/*kernel_stats.csv; do [ -f "$f" ] && head -5 "$f" | cut -c1-150; done
tail -1 $O/prof_run.log | head -c 200
true
