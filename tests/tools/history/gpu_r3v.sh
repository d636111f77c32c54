# round 3, call V: phase stamps of step_main_ws at the background shape, three-tile vs two-tile rounds; kernel trace of the new plan
set -x
mkdir -p gpurun_out/r3v
O=$PWD/gpurun_out/r3v
python tests/tools/phase_profile.py background split 0 > $O/phases_nt3.txt 2>&1; cat $O/phases_nt3.txt
python tests/tools/phase_profile.py background split 4 > $O/phases_nt2.txt 2>&1; cat $O/phases_nt2.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nt3 -o bg -- python $GRAFT_REPO_ROOT/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/prof_nt3.log 2>&1
find $O/prof_nt3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_nt3.csv
head -8 $O/kernel_stats_nt3.csv
rm -rf $O/prof_nt3
true
