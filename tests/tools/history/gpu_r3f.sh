# round 3, call F: phase clocks of the LAST round of every workgroup of step_main_ws: the tail round (one tile, adds into the
# workgroup's row) and, without tail rounds, the second full round
set -x
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3f
for f in 0 2 3; do
timeout 100 python tests/tools/phase_profile.py background split $f > $O/phases_background_flags$f.txt 2>&1; tail -19 $O/phases_background_flags$f.txt
done
true
