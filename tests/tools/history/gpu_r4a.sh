# round 3, call 4A: single-round specialisation of step_main_ws (hidden 128) + balanced encoding for three-tile rounds
set -x
mkdir -p gpurun_out/r4a
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -k "three_tile or full_size_properties_of_the_other or seeded_shapes or generic_width_kernel or shared_background or background_on_second or frame_trajectory or two_ranks" 2>&1 | tail -4 > $O/pytest_sel.txt; cat $O/pytest_sel.txt
cd /tmp
for c in background background_rank4 background_rank8; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o bg -- python $R/bench.py --config $c --steps 400 --warmup 40 --timed-only > $O/prof_$c.log 2>&1
find $O/prof_$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$c.csv; echo "== $c"; head -3 $O/kernel_stats_$c.csv | cut -c1-150; grep '"value"' $O/prof_$c.log | tail -1 | cut -c1-150; rm -rf $O/prof_$c
done
python $R/tests/tools/phase_profile.py background split 0 > $O/phases_nt3.txt 2>&1; head -20 $O/phases_nt3.txt
true
