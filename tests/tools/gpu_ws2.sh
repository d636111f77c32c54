set -x
mkdir -p gpurun_out/wp
export TMPDIR=/tmp
O=$PWD/gpurun_out/wp
R=$PWD
cd /tmp
for k in auto wp; do
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$k -o bg -- python $R/bench.py --config background --kernel $k --timed-only --steps 200 --warmup 20 > $O/prof_run_$k.log 2>&1 < /dev/null
for f in $O/prof_$k/*kernel_stats.csv; do [ -f "$f" ] && head -3 "$f" | cut -c1-140; done
done
