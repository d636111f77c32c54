#!/bin/bash
# gpu_bench_with_pmc.sh [config] [weights] [steps] [warmup] - run INSIDE one gpurun call:
#   1. the FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, one counter per pass, --kernel-trace only - as MI355X_MICROARCH.md's HBM
#      section prescribes) over tests/tools/run_steps.py of the configuration, folded by tests/tools/pmc_summary.py (which records
#      the SHA-256 of the library the passes ran);
#   2. bench.py --pmc-file <that summary>: roofline.traffic is then OBSERVED for the library being benchmarked (the line carries both
#      hashes and `same_library`), not copied from a committed file.
# Output: gpurun_out/bench_pmc/{pmc_counters.json, bench.json}.  
set -x
CFG=${1:-replica_room0_vmap}; WTS=${2:-f32}; STEPS=${3:-20}; WARM=${4:-5}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/bench_pmc
mkdir -p $O; rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$C -o p -- python $R/tests/tools/run_steps.py $CFG 40 $WTS > $O/pmc_$C.log 2>&1 < /dev/null
  echo "$C rc=$?"
done
cd $R
PMC_WORKLOAD="tests/tools/run_steps.py $CFG 40 $WTS" python tests/tools/pmc_summary.py > $O/pmc_counters.json
rm -rf gpurun_out/pmc
python bench.py --config $CFG --weights $WTS --steps $STEPS --warmup $WARM --pmc-file $O/pmc_counters.json > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
