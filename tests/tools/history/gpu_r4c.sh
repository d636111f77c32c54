# round 3, call 4C: three-tile rounds with both encoding jobs' points requested up front: phase stamps + kernel time
set -x
mkdir -p gpurun_out/r4c
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4c
python tests/tools/phase_profile.py background split 0 > $O/phases_nt3.txt 2>&1; head -6 $O/phases_nt3.txt
cd /tmp
for rep in 1 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bg -- python $R/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_background_$rep.csv; head -3 $O/kernel_stats_background_$rep.csv | cut -c1-150; grep '"value"' $O/prof.log | tail -1 | cut -c1-120; rm -rf $O/prof
done
true
