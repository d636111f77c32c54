# round 3, call Q: HBM counters of the background step (final library), its bench line with floor_us, phase clocks
set -x
mkdir -p gpurun_out/r3q
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3q
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc/$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$C -o p -- python $R/tests/tools/run_steps.py background 40 > $O/pmc_$C.log 2>&1 < /dev/null
done
cd $R
python tests/tools/pmc_summary.py > $O/pmc_counters_background.json 2>$O/pmc_summary.err; tail -9 $O/pmc_counters_background.json
