# round 3, call 4D: where the other configs spend their step: kernel traces of imap_plumbing (hidden 256), stress (hidden 64), scannet (50 objects)
set -x
mkdir -p gpurun_out/r4d
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4d
cd /tmp
for c in imap_plumbing stress_256x64 scannet0024_vmap; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bg -- python $R/bench.py --config $c --steps 100 --warmup 20 --timed-only > $O/prof_$c.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$c.csv; echo "== $c"; head -4 $O/kernel_stats_$c.csv | cut -c1-170; grep '"value"' $O/prof_$c.log | tail -1 | cut -c1-120; rm -rf $O/prof
done
true
